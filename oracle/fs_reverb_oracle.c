/*
 * fs_reverb_oracle.c — CPU restatement of the reverb plugin's per-callback path (row f2).  TEST INFRASTRUCTURE ONLY.
 *
 * Built by oracle/Makefile target `_ref` TOGETHER WITH the reference's own vendored KissFFT sources, compiled
 * where they lie under /root/reference (never copied into this repo):
 *   Plugins/FrequenSee/Source/FrequenSee/Private/FrequenSeeFFTConvolver/KissFFT/{kiss_fft.c,kiss_fftr.c}
 * so the FFTs below ARE the reference's FFTs.  Restated around them:
 *   FFrequenSeeAudioReverbPlugin::Initialize        RVB.cpp:74-102   (sizes, FFTSize = RoundUpToPowerOfTwo(47999+1024))
 *   FFrequenSeeAudioReverbPlugin::ProcessSourceAudio RVB.cpp:118-170
 *   FFrequenSeeAudioReverbPlugin::ConvolveFFT        RVB.cpp:172-213
 *   FCircularAudioBuffer                             CircularBuffer.cpp:15-74
 * (RVB.cpp = Plugins/FrequenSee/Source/FrequenSee/Private/FrequenSeeAudioReverbPlugin.cpp)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "kiss_fftr.h"

typedef struct ring { float* buf; int32_t size, head; } ring;   /* FCircularAudioBuffer */

static void ring_set_size(ring* r, int32_t n) { r->buf = (float*)calloc((size_t)n, sizeof(float)); r->size = n; r->head = 0; }
static void ring_add(ring* r, float s) { r->buf[r->head] = s; r->head = (r->head + 1) % r->size; }   /* CB.cpp:28-33 */
static void ring_last(const ring* r, float* out, int32_t count) {                                     /* CB.cpp:58-74 */
    int32_t start = (r->head - count + r->size) % r->size;
    for (int32_t i = 0; i < count; ++i) out[i] = r->buf[(start + i) % r->size];
}

typedef struct fso_reverb {
    int32_t ir_size, frame, fft_size, tail_size;
    ring tail_l, tail_r;
    float *cur_l, *cur_r, *conv_l, *conv_r, *ir_padded, *in_padded, *time_out;
    kiss_fft_cpx *in_f, *ir_f, *out_f;
    kiss_fftr_cfg fwd, inv;
} fso_reverb;

static int32_t round_up_pow2(int32_t v) { int32_t p = 1; while (p < v) p <<= 1; return p; }

fso_reverb* fso_reverb_create(int32_t sample_rate, float simulated_duration, int32_t frame_size) {   /* RVB.cpp:74-102 */
    fso_reverb* r = (fso_reverb*)calloc(1, sizeof(*r));
    r->ir_size = (int32_t)((float)sample_rate * simulated_duration);
    r->frame = frame_size;
    r->tail_size = r->ir_size - 1;
    ring_set_size(&r->tail_l, r->tail_size);
    ring_set_size(&r->tail_r, r->tail_size);
    int32_t cur = r->ir_size - 1 + frame_size;
    r->cur_l = (float*)calloc((size_t)cur, sizeof(float));
    r->cur_r = (float*)calloc((size_t)cur, sizeof(float));
    r->conv_l = (float*)calloc((size_t)cur, sizeof(float));
    r->conv_r = (float*)calloc((size_t)cur, sizeof(float));
    r->fft_size = round_up_pow2(cur);
    int32_t bins = r->fft_size / 2 + 1;
    r->fwd = kiss_fftr_alloc(r->fft_size, 0, NULL, NULL);
    r->inv = kiss_fftr_alloc(r->fft_size, 1, NULL, NULL);
    r->in_f = (kiss_fft_cpx*)calloc((size_t)bins, sizeof(kiss_fft_cpx));
    r->ir_f = (kiss_fft_cpx*)calloc((size_t)bins, sizeof(kiss_fft_cpx));
    r->out_f = (kiss_fft_cpx*)calloc((size_t)bins, sizeof(kiss_fft_cpx));
    r->time_out = (float*)calloc((size_t)r->fft_size, sizeof(float));
    r->ir_padded = (float*)calloc((size_t)r->fft_size, sizeof(float));
    r->in_padded = (float*)calloc((size_t)r->fft_size, sizeof(float));
    return r;
}

void fso_reverb_destroy(fso_reverb* r) {
    if (!r) return;
    free(r->tail_l.buf); free(r->tail_r.buf); free(r->cur_l); free(r->cur_r); free(r->conv_l); free(r->conv_r);
    free(r->in_f); free(r->ir_f); free(r->out_f); free(r->time_out); free(r->ir_padded); free(r->in_padded);
    kiss_fftr_free(r->fwd); kiss_fftr_free(r->inv);
    free(r);
}

int32_t fso_reverb_fft_size(const fso_reverb* r) { return r->fft_size; }

static void convolve_fft(fso_reverb* r, const float* ir, const float* input, float* output) {        /* RVB.cpp:172-213 */
    const int32_t ir_size = r->ir_size, in_size = r->tail_size + r->frame, conv_size = in_size;
    const int32_t bins = r->fft_size / 2 + 1;
    memcpy(r->in_padded, input, sizeof(float) * (size_t)in_size);
    memcpy(r->ir_padded, ir, sizeof(float) * (size_t)ir_size);
    kiss_fftr(r->fwd, r->in_padded, r->in_f);
    kiss_fftr(r->fwd, r->ir_padded, r->ir_f);
    for (int32_t i = 0; i < bins; ++i) {
        kiss_fft_cpx a = r->in_f[i], b = r->ir_f[i];
        r->out_f[i].r = a.r * b.r - a.i * b.i;
        r->out_f[i].i = a.r * b.i + a.i * b.r;
    }
    kiss_fftri(r->inv, r->out_f, r->time_out);
    const float scale = 1.0f / (float)r->fft_size;
    for (int32_t i = ir_size - 1; i < conv_size; ++i) r->time_out[i] *= scale;
    memcpy(output + (ir_size - 1), r->time_out + (ir_size - 1), sizeof(float) * (size_t)r->frame);
}

/* ProcessSourceAudio RVB.cpp:118-170.  in/out: interleaved stereo [frame * 2]; ir_l/ir_r: GetImpulseResponse()[0/1].
 * literal_tail != 0 reproduces RVB.cpp:147-148 (the first `frame` floats of the INTERLEAVED buffer are copied
 * into both mono tails); 0 = evident intent (each channel's own samples). */
void fso_reverb_process(fso_reverb* r, const float* ir_l, const float* ir_r, const float* in, float* out,
                        int32_t apply_reverb, int32_t literal_tail) {
    const int32_t frame = r->frame, tail = r->tail_size;
    if (!apply_reverb) { memcpy(out, in, sizeof(float) * (size_t)frame * 2); return; }   /* RVB.cpp:128-132 */
    ring_last(&r->tail_l, r->cur_l, tail);                                               /* RVB.cpp:141-142 */
    ring_last(&r->tail_r, r->cur_r, tail);
    for (int32_t i = 0; i < frame; ++i) ring_add(&r->tail_l, in[0 + i * 2]);            /* RVB.cpp:144-145 */
    for (int32_t i = 0; i < frame; ++i) ring_add(&r->tail_r, in[1 + i * 2]);
    if (literal_tail) {
        memcpy(r->cur_l + tail, in, sizeof(float) * (size_t)frame);                      /* RVB.cpp:147-148 */
        memcpy(r->cur_r + tail, in, sizeof(float) * (size_t)frame);
    } else {
        for (int32_t i = 0; i < frame; ++i) { r->cur_l[tail + i] = in[2 * i]; r->cur_r[tail + i] = in[2 * i + 1]; }
    }
    convolve_fft(r, ir_l, r->cur_l, r->conv_l);                                          /* RVB.cpp:151 */
    convolve_fft(r, ir_r, r->cur_r, r->conv_r);                                          /* RVB.cpp:154 */
    for (int32_t s = 0; s < frame; ++s) {                                                /* RVB.cpp:163-169, MixAlpha = 1 */
        float l = r->conv_l[tail + s], q = r->conv_r[tail + s];
        l = l < -1.0f ? -1.0f : (l > 1.0f ? 1.0f : l);
        q = q < -1.0f ? -1.0f : (q > 1.0f ? 1.0f : q);
        out[s * 2] = l * 1.0f + in[s * 2] * (1.0f - 1.0f);
        out[s * 2 + 1] = q * 1.0f + in[s * 2 + 1] * (1.0f - 1.0f);
    }
}

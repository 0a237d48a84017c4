/*
 * fs_material_oracle.c — CPU restatement of UMaterialAcousticProcessor::ApplyMaterialFD (row f4).
 * TEST INFRASTRUCTURE ONLY.
 *
 * Built by oracle/Makefile target `_ref` together with the reference's own vendored KissFFT (compiled where it
 * lies under /root/reference, never copied), so the transforms below ARE the reference's transforms.
 * Restated: Plugins/FrequenSee/Source/FrequenSee/Private/MaterialAcousticProcessor.cpp:8-107 (MAP.cpp)
 *   :13-17   N = next power of two >= L, NumBins = N/2 + 1
 *   :20-26   all three response curves must have NumBins entries, else an error is logged and the
 *            outputs are empty                                   -> return -1
 *   :29-30   zero-padded copy of the input
 *   :46-47   forward real FFT
 *   :51-72   per-bin gains: Refl = 1 - alpha; tau clamped so Refl + tau <= 1;
 *            specular Refl*(1 - sigma), diffuse Refl*sigma, transmitted tau
 *   :75-77   three inverse real FFTs
 *   :86-92   scale by 1/N, first L samples
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "kiss_fftr.h"

int32_t fso_material_fft_size(int32_t L) { int32_t n = 1; while (n < L) n <<= 1; return n; }

int32_t fso_apply_material_fd(const float* in, int32_t L, const float* absorption, const float* transmission,
                              const float* scattering, int32_t num_responses, float* specular, float* diffuse,
                              float* transmitted) {
    const int32_t N = fso_material_fft_size(L);
    const int32_t bins = N / 2 + 1;
    if (num_responses != bins) return -1;                                        /* MAP.cpp:20-26 */
    if (N < 2) {   /* kiss_fftr needs an even size; a 1-sample block is its own spectrum */
        if (L == 1) {
            float a = absorption[0], t = transmission[0], s = scattering[0];
            float refl = 1.0f - a;
            if (refl + t > 1.0f) t = 1.0f - refl;
            specular[0] = in[0] * (refl * (1.0f - s));
            diffuse[0] = in[0] * (refl * s);
            transmitted[0] = in[0] * t;
        }
        return 0;
    }
    float* time_in = (float*)calloc((size_t)N, sizeof(float));                    /* :29-30 */
    memcpy(time_in, in, sizeof(float) * (size_t)L);
    kiss_fft_cpx* f_in = (kiss_fft_cpx*)calloc((size_t)bins, sizeof(kiss_fft_cpx));
    kiss_fft_cpx* f_spec = (kiss_fft_cpx*)calloc((size_t)bins, sizeof(kiss_fft_cpx));
    kiss_fft_cpx* f_diff = (kiss_fft_cpx*)calloc((size_t)bins, sizeof(kiss_fft_cpx));
    kiss_fft_cpx* f_trans = (kiss_fft_cpx*)calloc((size_t)bins, sizeof(kiss_fft_cpx));
    float* t_spec = (float*)calloc((size_t)N, sizeof(float));
    float* t_diff = (float*)calloc((size_t)N, sizeof(float));
    float* t_trans = (float*)calloc((size_t)N, sizeof(float));
    kiss_fftr_cfg fwd = kiss_fftr_alloc(N, 0, NULL, NULL);                        /* :42-43 */
    kiss_fftr_cfg inv = kiss_fftr_alloc(N, 1, NULL, NULL);
    kiss_fftr(fwd, time_in, f_in);                                                /* :46 */
    for (int32_t b = 0; b < bins; ++b) {                                          /* :51-72 */
        const float alpha = absorption[b];
        float tau = transmission[b];
        const float sigma = scattering[b];
        const float refl = 1.0f - alpha;
        if (refl + tau > 1.0f) tau = 1.0f - refl;
        const float g_spec = refl * (1.0f - sigma);
        const float g_diff = refl * sigma;
        const float g_trans = tau;
        f_spec[b].r = f_in[b].r * g_spec;   f_spec[b].i = f_in[b].i * g_spec;
        f_diff[b].r = f_in[b].r * g_diff;   f_diff[b].i = f_in[b].i * g_diff;
        f_trans[b].r = f_in[b].r * g_trans; f_trans[b].i = f_in[b].i * g_trans;
    }
    kiss_fftri(inv, f_spec, t_spec);                                              /* :75-77 */
    kiss_fftri(inv, f_diff, t_diff);
    kiss_fftri(inv, f_trans, t_trans);
    const float scale = 1.0f / (float)N;                                          /* :86-92 */
    for (int32_t i = 0; i < L; ++i) {
        specular[i] = t_spec[i] * scale;
        diffuse[i] = t_diff[i] * scale;
        transmitted[i] = t_trans[i] * scale;
    }
    free(fwd); free(inv);
    free(time_in); free(f_in); free(f_spec); free(f_diff); free(f_trans); free(t_spec); free(t_diff); free(t_trans);
    return 0;
}

/* oracle_san.c — drives every entry point of the CPU oracle once under AddressSanitizer + UBSan
 * (tests/test_native_sanitizers.py compiles this together with oracle/fs_oracle.c). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/fs_oracle.h"

int main(void) {
    /* shoebox 1000 x 800 x 300 + a wall */
    const float W = 1000, D = 800, H = 300;
    const float p[8][3] = {{0, 0, 0}, {W, 0, 0}, {W, D, 0}, {0, D, 0}, {0, 0, H}, {W, 0, H}, {W, D, H}, {0, D, H}};
    const int q[7][4] = {{0, 1, 2, 3}, {4, 5, 6, 7}, {0, 1, 5, 4}, {3, 2, 6, 7}, {0, 3, 7, 4}, {1, 2, 6, 5}, {0, 0, 0, 0}};
    float xyz[14 * 9]; uint16_t mat[14]; uint32_t obj[14]; int T = 0;
    for (int f = 0; f < 6; ++f) {
        const int tri[2][3] = {{q[f][0], q[f][1], q[f][2]}, {q[f][0], q[f][2], q[f][3]}};
        for (int t = 0; t < 2; ++t, ++T) { for (int v = 0; v < 3; ++v) memcpy(&xyz[T * 9 + v * 3], p[tri[t][v]], 12); mat[T] = (uint16_t)(f % 2); obj[T] = 0; }
    }
    const float wall[2][9] = {{500, 0, 0, 500, 800, 0, 500, 800, 150}, {500, 0, 0, 500, 800, 150, 500, 0, 150}};
    for (int t = 0; t < 2; ++t, ++T) { memcpy(&xyz[T * 9], wall[t], 36); mat[T] = 0xFFFFu; obj[T] = 5; }
    const float absorption[2 * 3] = {0.5f, 0.4f, 0.3f, 0.8f, 0.1f, 0.05f};
    fso_scene* s = fso_scene_create(xyz, mat, T, absorption, 2, 3);
    fso_scene_set_objects(s, obj);
    fso_params prm; fso_params_default(&prm); prm.num_pairs = 600; prm.depth = 0;
    const float src[3] = {250, 200, 150}, lis[3] = {750, 600, 120};
    float* e32 = (float*)malloc(sizeof(float) * 3 * 1000); double* e64 = (double*)malloc(sizeof(double) * 3 * 1000);
    fso_counters c; memset(&c, 0, sizeof(c));
    fso_compute_energy(s, &prm, src, lis, 0, 600, 1000, e32, e64, &c);
    prm.flags = FSO_FLAG_BRUTE_FORCE | FSO_FLAG_COSINE_SAMPLING | FSO_FLAG_FIXED_NORM_1000; prm.depth = 5;
    fso_compute_energy(s, &prm, src, lis, 100, 200, 1000, e32, NULL, &c);
    float* ir = (float*)malloc(sizeof(float) * 48000);
    fso_reconstruct(e32, 1000, 48000, 0.001f, 48000, 0, ir);
    fso_reconstruct(e32, 1000, 48000, 0.001f, 48000, 48, ir);
    fso_sound_params sp; fso_sound_params_default(&sp); sp.raycasts_per_tick = 300;
    fso_sound_result sr; fso_update_sound(s, &sp, src, lis, &sr);
    fso_save_array_to_file(ir, 48000, "/tmp/fs_oracle_san_ir.txt");
    int n = fso_load_float_array("/tmp/fs_oracle_san_ir.txt", ir, 48000);
    /* empty scene */
    fso_scene* e = fso_scene_create(NULL, NULL, 0, NULL, 0, 1);
    fso_params_default(&prm); prm.num_pairs = 16; prm.depth = 3;
    fso_compute_energy(e, &prm, src, lis, 0, 16, 1000, e32, NULL, &c);
    fso_scene_destroy(e);
    printf("connected=%llu occlusion=%g lines=%d\n", (unsigned long long)c.connected, sr.occlusion_attenuation, n);
    free(e32); free(e64); free(ir); fso_scene_destroy(s);
    return n == 48000 ? 0 : 1;
}

// fs_harness.cpp — headless C++ harness: the host side a game engine would own, driving the HIP path
// through the C ABI only (no HIP headers, no Python).  Mirrors the reference's per-frame sequence:
//   RegisterGeometry / RegisterSource -> every frame: UpdateSource (trace + deposit + reconstruct) ->
//   audio thread reads GetImpulseResponse(); optionally UpdateSound() for the occlusion scalar and
//   SaveArrayToFile("saved_ir.txt").
//
//   usage: fs_harness [frames=100] [pairs=1000] [depth=0] [out=saved_ir.txt]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "FrequenSee.hpp"

using namespace frequensee;

static void add_quad(AcousticGeometryComponent& g, const float a[3], const float b[3], const float c[3], const float d[3]) {
    const float* t[6] = {a, b, c, a, c, d};
    for (int i = 0; i < 6; ++i) for (int k = 0; k < 3; ++k) g.Triangles.push_back(t[i][k]);
    g.MaterialId.push_back(0); g.MaterialId.push_back(0);
}

int main(int argc, char** argv) {
    const int frames = argc > 1 ? std::atoi(argv[1]) : 100;
    const int pairs = argc > 2 ? std::atoi(argv[2]) : AudioRayTracingSubsystem::USED_RAY_COUNT;
    const int depth = argc > 3 ? std::atoi(argv[3]) : 0;
    const std::string out = argc > 4 ? argv[4] : "saved_ir.txt";
    const int pipelining = argc > 5 ? std::atoi(argv[5]) : 0;   // fs_set_pipelining depth: Tick then streams the sources
    try {
        AudioRayTracingSubsystem SubSys(/*NumBands=*/1);
        // shoebox 1000 x 800 x 300 cm, one material rho = 0.5 (BASELINE.json configs[0])
        AcousticGeometryComponent Room;
        const float W = 1000, D = 800, H = 300;
        const float p[8][3] = {{0, 0, 0}, {W, 0, 0}, {W, D, 0}, {0, D, 0}, {0, 0, H}, {W, 0, H}, {W, D, H}, {0, D, H}};
        add_quad(Room, p[0], p[1], p[2], p[3]); add_quad(Room, p[4], p[5], p[6], p[7]);
        add_quad(Room, p[0], p[1], p[5], p[4]); add_quad(Room, p[3], p[2], p[6], p[7]);
        add_quad(Room, p[0], p[3], p[7], p[4]); add_quad(Room, p[1], p[2], p[6], p[5]);
        SubSys.RegisterGeometry(&Room);
        SubSys.SetMaterials({0.5f}, 1);
        FrequenSeeAudioComponent Comp({250, 200, 150});
        Comp.OnRegister(SubSys);
        SubSys.SetListenerLocation({750, 600, 120});
        SubSys.Params.num_rays = 2u * (uint32_t)pairs;
        SubSys.Params.depth = depth;

        std::vector<float> Energy;
        SubSys.UpdateSource(Comp, &Energy);                       // warm-up + first result
        if (pipelining) SubSys.SetPipelining(pipelining);
        auto t0 = std::chrono::steady_clock::now();
        for (int f = 0; f < frames; ++f) {
            SubSys.Params.seed = 0x5EED + (uint64_t)f;            // a new sample set every frame
            SubSys.Tick(1.0f / 60.0f);
        }
        SubSys.Synchronize();                                     // (streamed Ticks only submit)
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        int n = 0;
        const float* ir = Comp.GetImpulseResponse(0, &n);
        double e = 0, peak = 0;
        for (float v : Energy) e += v;
        for (int i = 0; i < n; ++i) peak = std::fmax(peak, std::fabs(ir[i]));
        fs_sound_result snd = Comp.UpdateSound();
        Comp.SaveImpulseResponse(out);
        // ApplyMaterialFD on the traced IR: a half-absorbing, fully diffuse wall leaves 0.5 x the block in Diffuse
        MaterialAcousticProcessor Proc(SubSys);
        std::vector<float> Block(ir, ir + n);
        int fft = 1;
        while (fft < n) fft <<= 1;
        MaterialAcousticFD Props;
        Props.Absorption.assign((size_t)fft / 2 + 1, 0.5f);
        Props.Transmission.assign((size_t)fft / 2 + 1, 0.0f);
        Props.Scattering.assign((size_t)fft / 2 + 1, 1.0f);
        AcousticOutputs Fd = Proc.ApplyMaterialFD(Block, Props);
        double fd_err = 0;
        for (int i = 0; i < n; ++i) fd_err = std::fmax(fd_err, std::fabs(Fd.Diffuse[(size_t)i] - 0.5 * ir[i]));
        std::printf("{\"frames\": %d, \"pairs\": %d, \"depth\": %d, \"frames_per_s\": %.1f, \"rays_per_s\": %.0f, "
                    "\"energy_sum_first_frame\": %.6f, \"ir_samples\": %d, \"ir_peak\": %.6f, "
                    "\"occlusion_attenuation\": %.6f, \"legacy_rays_reaching\": %u, \"material_fd_max_err\": %.3g, "
                    "\"saved\": \"%s\"}\n",
                    frames, pairs, depth, frames / s, 2.0 * pairs * frames / s, e, n, peak,
                    snd.occlusion_attenuation, snd.rays_reaching_listener, fd_err, out.c_str());
    } catch (const std::exception& ex) {
        std::fprintf(stderr, "fs_harness: %s\n", ex.what());
        return 1;
    }
    return 0;
}

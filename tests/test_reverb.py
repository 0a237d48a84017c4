"""Row f2: the reverb plugin's per-callback convolution (ProcessSourceAudio RVB.cpp:118-170, ConvolveFFT
:172-213).  The oracle (oracle/_ref) is built around the REFERENCE'S OWN KissFFT sources; the HIP path
evaluates the same output samples as a direct convolution."""
import numpy as np
import pytest

FRAME = 1024


def blocks(rng, n):
    return [np.clip(rng.normal(0, 0.3, 2 * FRAME), -1, 1).astype(np.float32) for _ in range(n)]


def test_reference_kissfft_oracle_kats(oracle_mod):
    ref = oracle_mod.ReverbRef()
    assert ref.fft_size == 65536                       # RoundUpToPowerOfTwo(47999 + 1024), A.7-5
    rng = np.random.default_rng(0)
    delta = np.zeros(48000, np.float32)
    delta[0] = 1.0
    x = blocks(rng, 3)
    for b in x:                                        # A.7-6: IR = delta[0] => output block = input block
        y = ref.process(delta, delta, b)
        assert np.abs(y - b).max() < 2e-6
    # literal HEAD behaviour (RVB.cpp:147-148): the interleaved buffer's first FRAME floats feed BOTH channels
    ref2 = oracle_mod.ReverbRef()
    y = ref2.process(delta, delta, x[0], literal_tail=True)
    assert np.abs(y[0::2] - x[0][:FRAME]).max() < 2e-6 and np.abs(y[1::2] - x[0][:FRAME]).max() < 2e-6
    # bApplyReverb == false: bypass
    assert np.array_equal(ref2.process(delta, delta, x[1], apply_reverb=False), x[1])
    # a delayed, scaled tap and the clamp: IR = 0.5 delta[FRAME] reproduces the previous block at half gain
    ref3 = oracle_mod.ReverbRef()
    ir = np.zeros(48000, np.float32)
    ir[FRAME] = 0.5
    ref3.process(ir, ir, x[0])
    y = ref3.process(ir, ir, x[1])
    assert np.abs(y - 0.5 * x[0]).max() < 2e-6
    big = np.zeros(48000, np.float32)
    big[0] = 3.0
    assert np.abs(ref3.process(big, big, x[2])).max() <= 1.0     # FMath::Clamp(.., -1, 1)


def test_fft_product_equals_direct_convolution(oracle_mod):
    """For the kept samples [47999, 49023) the 65536-point circular product IS the plain convolution."""
    ref = oracle_mod.ReverbRef()
    rng = np.random.default_rng(1)
    ir = (rng.normal(0, 1, 48000) * np.exp(-np.arange(48000) / 6000.0) * 0.02).astype(np.float32)
    hist = np.zeros((2, 47999 + FRAME), np.float64)
    for b in blocks(rng, 4):
        y = ref.process(ir, ir, b)
        for ch in range(2):
            hist[ch] = np.concatenate([hist[ch][FRAME:], b[ch::2].astype(np.float64)])
            want = np.array([np.dot(ir.astype(np.float64), hist[ch][s:s + 48000][::-1]) for s in range(FRAME)])
            assert np.abs(y[ch::2] - np.clip(want, -1, 1)).max() < 5e-5 * max(1.0, np.abs(want).max())


@pytest.mark.gpu
def test_gpu_reverb_matches_reference_kissfft(pkg, oracle_mod, scene_factory):
    sc = scene_factory("starter_room", 4)
    sub = pkg.AudioRayTracingSubsystem(num_bands=4)
    sub.RegisterGeometry(sc.triangles, sc.material_ids)
    sub.SetMaterials(sc.absorption)
    comp = pkg.FrequenSeeAudioComponent(sc.source)
    comp.OnRegister(sub)
    sub.SetListenerLocation(sc.listener)
    sub.UpdateSource(comp, pkg.default_params(num_rays=16384, depth=8, dist_divisor=100.0))   # a real traced IR
    ir = [v.copy() for v in comp.GetImpulseResponse()]
    assert ir[0].any()
    plug = pkg.FrequenSeeAudioReverbPlugin(sub)
    plug.Initialize(BufferLength=FRAME)
    plug.OnInitSource(comp)
    rng = np.random.default_rng(2)
    for literal in (False, True):
        ref = oracle_mod.ReverbRef()
        plug.OnReleaseSource(comp)                          # ClearBuffers
        for b in blocks(rng, 5):
            got = plug.ProcessSourceAudio(comp, b, literal_tail=literal)
            want = ref.process(ir[0], ir[1], b, literal_tail=literal)
            assert np.abs(got - want).max() <= 2e-5 * max(np.abs(want).max(), 1e-3), literal
    comp.bApplyReverb = False                               # bypass (RVB.cpp:128-132)
    b = blocks(rng, 1)[0]
    assert np.array_equal(plug.ProcessSourceAudio(comp, b), b)
    sub.Deinitialize()


@pytest.mark.gpu
def test_gpu_reverb_with_installed_impulse_response(pkg, oracle_mod, tmp_path):
    """The authors' convolver check (README "works with a known IR"): GenerateDummyImpulseResponse
    (FSAC.cpp:408-452: a delta at samples 0 and N-1) and an IR loaded from a one-float-per-line text file
    (LoadFloatArray :454-490), installed through the mutable GetImpulseResponse() reference (FSAC.h:113)."""
    sub = pkg.AudioRayTracingSubsystem(num_bands=2)
    comp = pkg.FrequenSeeAudioComponent((0.0, 0.0, 0.0))
    comp.OnRegister(sub)
    plug = pkg.FrequenSeeAudioReverbPlugin(sub)
    plug.Initialize(BufferLength=FRAME)
    plug.OnInitSource(comp)
    rng = np.random.default_rng(5)
    dummy = comp.GenerateDummyImpulseResponse()
    assert dummy[0] == 1.0 and dummy[-1] == 1.0 and dummy.sum() == 2.0
    got_ir = comp.GetImpulseResponse()
    assert np.array_equal(got_ir[0], dummy) and np.array_equal(got_ir[1], dummy)
    assert np.array_equal(comp.GetBandImpulseResponse(1), dummy)
    ref = oracle_mod.ReverbRef()
    for b in blocks(rng, 4):
        want = ref.process(dummy, dummy, b)
        got = plug.ProcessSourceAudio(comp, b)
        assert np.abs(got - want).max() <= 2e-5 * max(np.abs(want).max(), 1e-3)
    # an exponentially decaying noise IR written and read back as text (six decimals, FSAC.cpp:492-505)
    ir = (rng.normal(0, 1, 48000) * np.exp(-np.arange(48000) / 5000.0) * 0.05).astype(np.float32)
    path = tmp_path / "loaded_ir.txt"
    pkg._capi.save_array_to_file(ir, path)
    loaded = pkg._capi.load_float_array(path)
    comp.SetImpulseResponse(loaded)
    plug.OnReleaseSource(comp)
    ref = oracle_mod.ReverbRef()
    for b in blocks(rng, 3):
        want = ref.process(loaded, loaded, b)
        got = plug.ProcessSourceAudio(comp, b)
        assert np.abs(got - want).max() <= 2e-5 * max(np.abs(want).max(), 1e-3)
    with pytest.raises(pkg.FrequenSeeError):
        comp.SetImpulseResponse(np.zeros(100, np.float32))           # wrong length
    sub.Deinitialize()


@pytest.mark.gpu
def test_offline_audition_tool(pkg, tmp_path):
    """tools/convolve_ir.py: saved_ir.txt + input_audio.txt -> output_audio.txt (the reference's bGenerateReverb
    round trip, FSAC.cpp:300-305) equals numpy's convolution, clamped to [-1, 1] like the plugin output."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("convolve_ir", os.path.join(root, "tools", "convolve_ir.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    rng = np.random.default_rng(11)
    ir = (rng.normal(0, 1, 6000) * np.exp(-np.arange(6000) / 900.0) * 0.05).astype(np.float32)
    audio = np.clip(rng.normal(0, 0.2, 5000), -1, 1).astype(np.float32)
    pkg._capi.save_array_to_file(ir, tmp_path / "saved_ir.txt")
    pkg._capi.save_array_to_file(audio, tmp_path / "input_audio.txt")
    assert tool.main(["convolve_ir.py", str(tmp_path / "saved_ir.txt"), str(tmp_path / "input_audio.txt"),
                      str(tmp_path / "output_audio.txt")]) == 0
    got = pkg._capi.load_float_array(tmp_path / "output_audio.txt")
    ir_txt = pkg._capi.load_float_array(tmp_path / "saved_ir.txt")          # six decimals, as the tool saw them
    audio_txt = pkg._capi.load_float_array(tmp_path / "input_audio.txt")
    want = np.clip(np.convolve(audio_txt.astype(np.float64), ir_txt.astype(np.float64)), -1, 1)
    assert got.size == audio.size + 48000 - 1
    assert np.abs(got[: want.size] - want).max() <= 2e-5 * max(np.abs(want).max(), 1e-3) + 1e-6   # + text rounding
    assert np.abs(got[want.size:]).max() <= 1e-6

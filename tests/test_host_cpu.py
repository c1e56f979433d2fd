"""CPU tests (no GPU): the C-ABI library loads and exports every symbol that
include/vaenmf.h declares, host logic (sharding, statistics, geometry, model containers,
pickling, the M1 count quirk), and the drop-in import paths."""
import os
import pickle
import re

import numpy as np
import pytest
import torch

import vaenmf_oracle as orc
from helpers import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import vaenmf
    from vaenmf import _lib
    hdr = open(os.path.join(ROOT, "include", "vaenmf.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(vaenmf_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = _lib.lib()                       # loads libvaenmf.so; AttributeError if a bound symbol is missing
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.vaenmf_last_error() is not None


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vaenmf.engine import BatchEngine
    p = orc.xavier_normal_params([65, 32, [128, 128]], seed=0)
    dec = [p["decoder.hidden.0.weight"], p["decoder.hidden.0.bias"], p["decoder.hidden.1.weight"],
           p["decoder.hidden.1.bias"], p["decoder.reconstruction.weight"], p["decoder.reconstruction.bias"]]
    with pytest.raises(RuntimeError):
        BatchEngine(65, 4, dec)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "guided-vae-nmf_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "vaenmf_oracle" not in txt and "oracle/" not in txt, os.path.join(dirpath, f)


def test_stft_geometry_host():
    """stft.py:37-38 (integer window) and :48-53 (end-pad rule) on the host side of the C ABI."""
    from vaenmf.stft import frame_geometry
    assert frame_geometry(64000, 16000, 32e-3, 0.25) == (512, 128, 501, 64000)
    nfft, hop, nfr, npad = frame_geometry(63000, 16000, 64e-3, 0.25)
    X = orc.stft(np.zeros(63000), fs=16000, wlen_sec=64e-3)
    assert (nfft, hop) == (1024, 256) and nfr == X.shape[1] and npad == 63000 + 256
    for T in (16000, 16001, 40000, 64000, 70001, 99999):
        for wl in (32e-3, 64e-3):
            assert frame_geometry(T, 16000, wl, 0.25)[2] == orc.stft(np.zeros(T), fs=16000, wlen_sec=wl).shape[1]
    with pytest.raises(ValueError):
        frame_geometry(64000, 16000, 50.01e-3, 0.25)


def test_shard_is_array_split():
    """scripts/evaluate_M1.py:203: np.array_split(file_paths, nb_devices)."""
    from vaenmf.pipeline import shard
    files = ["f%03d" % i for i in range(27)]
    for ws in (1, 2, 4, 8):
        parts = [shard(files, ws, r) for r in range(ws)]
        ref = [list(a) for a in np.array_split(files, ws)]
        assert parts == ref and sum(parts, []) == files


def test_metric_statistics():
    """metrics.py:5-10, 70-108 from sufficient statistics == from the raw lists."""
    from vaenmf import metrics as vm
    g = np.random.default_rng(0)
    vals = g.normal(3, 2, (50, 3))
    snr = g.choice([-5.0, 0.0, 5.0], 50)
    st = vm.sufficient_stats(vals, snr)
    # splitting over two "ranks" and summing == all at once
    st2 = vm.sufficient_stats(vals[:20], snr[:20]) + vm.sufficient_stats(vals[20:], snr[20:])
    assert np.allclose(st, st2)
    tab = vm.stats_table(st)
    for k, key in enumerate(vm.METRIC_KEYS):
        m, h = orc.mean_confidence_interval(vals[:, k])
        assert tab[("all", key)][:2] == (m, h)
        for b in (-5.0, 0.0, 5.0):
            m, h = orc.mean_confidence_interval(vals[snr == b, k])
            assert tab[("snr=%g" % b, key)][:2] == (m, h)
    assert vm.mean_confidence_interval(vals[:, 0]) == orc.mean_confidence_interval(vals[:, 0])


def test_ratios_from_gram_match_reference_formula():
    from vaenmf import metrics as vm
    z = np.load(GOLDEN + "/metrics_dummy_m2.npz")
    sh, s, n = z["a_s_est"] / 32768.0, z["a_s"] / 32768.0, z["a_n"] / 32768.0
    G = np.array([sh @ sh, sh @ s, sh @ n, s @ s, s @ n, n @ n])
    assert np.allclose(vm.ratios_from_gram(G), z["a_ratios"], atol=1e-9)


def test_synth_generators_equal_oracle_copies():
    from vaenmf import synth
    a, b = synth.synth_utterance(3, 8000), orc.synth_utterance(3, 8000)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    pa = synth.xavier_normal_params([65, 32, [128, 128]], seed=2, y_dim=1, bias_std=0.1)
    pb = orc.xavier_normal_params([65, 32, [128, 128]], seed=2, y_dim=1, bias_std=0.1)
    assert pa.keys() == pb.keys() and all(np.array_equal(pa[k], pb[k]) for k in pa)


def test_model_containers_state_dict_layout():
    """Key layout of SURVEY 5 / models.py so reference checkpoints load unchanged."""
    import vaenmf
    vae = vaenmf.VariationalAutoencoder([513, 32, [128, 128]])
    keys = set(vae.state_dict())
    assert keys == {"encoder.hidden.0.weight", "encoder.hidden.0.bias", "encoder.hidden.1.weight", "encoder.hidden.1.bias",
                    "encoder.sample.mu.weight", "encoder.sample.mu.bias", "encoder.sample.log_var.weight",
                    "encoder.sample.log_var.bias", "decoder.hidden.0.weight", "decoder.hidden.0.bias",
                    "decoder.hidden.1.weight", "decoder.hidden.1.bias", "decoder.reconstruction.weight",
                    "decoder.reconstruction.bias"}
    assert vae.z_dim == 32 and float(vae.decoder.reconstruction.bias.abs().max()) == 0
    dgm = vaenmf.DeepGenerativeModel([513, 1, 32, [128, 128]], None)
    assert dgm.encoder.hidden[0].weight.shape == (128, 514) and dgm.decoder.hidden[0].weight.shape == (128, 33)
    clf = vaenmf.Classifier([513, [128, 128], 1])
    assert set(clf.state_dict()) == {"hidden.0.weight", "hidden.0.bias", "hidden.1.weight", "hidden.1.bias",
                                     "output_layer.weight", "output_layer.bias"}
    # forward == oracle on the same weights (plain torch forward of the container)
    p = orc.xavier_normal_params([65, 32, [128, 128]], seed=1, bias_std=0.1)
    v = vaenmf.VariationalAutoencoder([65, 32, [128, 128]])
    v.load_state_dict({k: torch.tensor(x) for k, x in p.items()})
    zin = np.random.default_rng(0).standard_normal((7, 32)).astype(np.float32)
    with torch.no_grad():
        out = v.decoder(torch.tensor(zin)).numpy()
    assert np.max(np.abs(out / orc.decoder_forward(p, zin) - 1)) < 1e-5


def test_mcem_objects_counts_and_pickle():
    import vaenmf
    m1 = vaenmf.MCEM_M1(niter=100)
    assert (m1.e_step_counts(), m1.wf_counts()) == ((30, 30), (75, 30))       # the positional shift, mcem.py:461-462
    q = np.load(GOLDEN + "/quirk_counts.npz")
    (r, b), (rw, bw) = m1.e_step_counts(), m1.wf_counts()
    assert (r + b, r, rw + bw, rw) == tuple(q["M1"])
    m2 = vaenmf.MCEM_M2(niter=100)
    (r, b), (rw, bw) = m2.e_step_counts(), m2.wf_counts()
    assert (r + b, r, rw + bw, rw) == tuple(q["M2"])
    assert vaenmf.MCEM_M1(niter=1, reference_compat=False).e_step_counts() == (10, 30)
    m = pickle.loads(pickle.dumps(m1))                                          # spawn-Pool transport, evaluate_M1.py:206-216
    assert m.niter == 100 and m.burnin_WF == 75


def test_dropin_import_paths():
    import importlib
    import sys
    pkg = os.path.join(ROOT, "guided-vae-nmf_amd")
    assert pkg in sys.path
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "python" or k.startswith("python.")}
    try:
        mc = importlib.import_module("python.models.mcem")
        md = importlib.import_module("python.models.models")
        st = importlib.import_module("python.processing.stft")
        me = importlib.import_module("python.metrics")
        tg = importlib.import_module("python.processing.target")
        sp = importlib.import_module("python.models.spp_estimation")
        assert sp.timo_mask_estimation and sp.timo_vad_estimation and sp.timo_noise_estimation
        assert mc.MCEM_M1.__name__ == "MCEM_M1" and md.VariationalAutoencoder and st.stft and me.energy_ratios
        assert mc.MCEM_M2_noNMF and tg.clean_speech_IBM and tg.clean_speech_VAD and tg.noise_robust_clean_speech_IBM
    finally:
        for k in list(sys.modules):
            if k == "python" or k.startswith("python."):
                del sys.modules[k]
        sys.modules.update(saved)


def test_wav_io_and_speech_list(tmp_path):
    """sf.read/sf.write semantics on RIFF PCM-16 (evaluate_M1.py:114,165) and the sorted recursive file list
    (python/dataset/csr1_wjs0_dataset.py:19-54)."""
    from vaenmf import wavio
    from vaenmf.driver import speech_list
    z = np.load(GOLDEN + "/metrics_dummy_m2.npz")
    pcm = z["a_s"][:5000]
    root = str(tmp_path) + "/"
    d = root + "CSR-1-WSJ-0/WAV/wsj0/si_et_05/440/"
    os.makedirs(d)
    os.makedirs(root + "CSR-1-WSJ-0/WAV/wsj0/si_tr_s/011/")
    for name in ("440c020b", "440c020a"):
        wavio.write(d + name + ".wav", pcm / 32768.0, 16000)
    wavio.write(root + "CSR-1-WSJ-0/WAV/wsj0/si_tr_s/011/011a010a.wav", pcm / 32768.0, 16000)
    x, fs = wavio.read(d + "440c020a.wav")
    assert fs == 16000 and x.dtype == np.float64 and len(x) == len(pcm)
    assert np.max(np.abs(x * 32768.0 - pcm)) <= 1.0          # one LSB: write scales by 32767, read by 1/32768
    assert speech_list(root, "test") == ["CSR-1-WSJ-0/WAV/wsj0/si_et_05/440/440c020a.wav",
                                         "CSR-1-WSJ-0/WAV/wsj0/si_et_05/440/440c020b.wav"]
    assert speech_list(root, "train") == ["CSR-1-WSJ-0/WAV/wsj0/si_tr_s/011/011a010a.wav"]
    wavio.write(root + "clip.wav", np.array([2.0, -2.0, 0.5]), 16000)
    assert list((wavio.read(root + "clip.wav")[0] * 32768).astype(int)) == [32767, -32768, 16384]


def test_nist_sphere_reader(tmp_path):
    """Raw WSJ0 files are NIST SPHERE (1024-byte ASCII header, then 16-bit PCM): the header layout is the one of the
    reference-committed data/subset/raw/.../440c020a.wav (sample_count 138583, sample_min -7385, sample_max 4084,
    sample_byte_format 01, sample_coding pcm -- checked against that file when the fixture was written); both byte
    orders, refusal of compressed payloads."""
    from vaenmf import wavio
    z = np.load(GOLDEN + "/metrics_dummy_m2.npz")
    pcm = z["a_s"][:7001].astype(np.int16)

    def sphere(order, coding="pcm"):
        lines = ["NIST_1A", "   1024", "database_id -s4 wsj0", "channel_count -i 1", "sample_count -i %d" % len(pcm),
                 "sample_min -i %d" % pcm.min(), "sample_max -i %d" % pcm.max(), "sample_rate -i 16000", "sample_n_bytes -i 2",
                 "sample_byte_format -s2 %s" % order, "sample_sig_bits -i 16", "sample_coding -s%d %s" % (len(coding), coding), "end_head"]
        hdr = ("\n".join(lines) + "\n").encode("ascii")
        return hdr + b" " * (1024 - len(hdr)) + pcm.astype("<i2" if order == "01" else ">i2").tobytes()

    for order in ("01", "10"):
        p = str(tmp_path / ("s%s.wav" % order))
        open(p, "wb").write(sphere(order))
        x, fs = wavio.read(p)
        assert fs == 16000 and x.dtype == np.float64 and np.array_equal(x * 32768.0, pcm.astype(np.float64))
    p = str(tmp_path / "shorten.wav")
    open(p, "wb").write(sphere("01", "pcm,embedded-shorten-v2.00"))
    with pytest.raises(NotImplementedError):
        wavio.read(p)


def test_wave_chain_addressability_guard():
    """The wave-private chain kernel uses 32-bit byte offsets per lane (idle lanes: 0xF0000000); a batch whose Zs / replay
    draws / spectrogram would pass them must run the 64-bit team kernel instead (vaenmf_mh_chain checks this function).
    Pure host arithmetic: callable without a GPU."""
    from vaenmf import _lib
    f = _lib.lib().vaenmf_wchain_addressable
    assert f(32064, 75, 105, 272, 8, 64, 0) == 1                        # BASELINE config 2
    assert f(125 * 251, 75, 105, 528, 32, 125, 0) == 1                  # config 5 shard
    # Zs [NT][Rcap][32] float: 0xE0000000 / (75 * 128) = 391 468 frames
    assert f(391_000, 75, 105, 272, 8, 700, 0) == 1 and f(392_000, 75, 105, 272, 8, 700, 0) == 0
    # replay draws [S][NT][32] float: 0xE0000000 / (105 * 128) = 279 620 frames
    assert f(279_000, 75, 105, 272, 8, 500, 1) == 1 and f(280_000, 75, 105, 272, 8, 500, 1) == 0
    assert f(280_000, 75, 105, 272, 8, 500, 0) == 1                     # the device generator needs no draw buffer
    # spectrogram rows [NT][Fs] float at the widest supported row
    assert f(1_400_000, 1, 2, 640, 8, 2000, 0) == 1 and f(1_500_000, 1, 2, 640, 8, 2000, 0) == 0


def test_committed_traffic_profiles_name_every_kernel_the_bench_reports():
    """bench.py reads the HBM bytes per launch of the dominant kernels from the committed rocprofv3 PMC summaries
    (profiles/round*_<tag>_traffic.json).  A renamed kernel must fail HERE, not silently drop a field of the bench line:
    every workload the bench reports has a round-3 profile whose kernels include the chain, the W statistics, the W
    update and the H/g kernel."""
    import json
    prof = os.path.join(ROOT, "profiles")
    need = {"bf16": ("chain_kernel", "wstats_fused", "w_update_tiles", "hg_stream"),
            "bf16x3": ("chain_kernel", "wstats_stream", "w_update", "hg_stream"),
            "bf16_M2ibm_f257_k8": ("chain_kernel", "wstats_fused", "w_update_tiles", "hg_stream"),
            "bf16_M2vad_f257_k8": ("chain_kernel", "wstats_fused", "w_update_tiles", "hg_stream"),
            "bf16_M1_f513_k10": ("chain_kernel", "wstats_stream", "w_update", "hg_stream"),
            "bf16_M1_f513_k32": ("chain_kernel", "wstats_stream", "w_update", "hg_stream")}
    for tag, subs in need.items():
        fn = os.path.join(prof, "round3_%s_traffic.json" % tag)
        assert os.path.exists(fn), fn
        kern = json.load(open(fn))["kernels"]
        for sub in subs:
            hit = [v for k, v in kern.items() if sub in k]
            assert hit and hit[0]["hbm_bytes_per_launch"] > 0 and hit[0]["launches"] > 0, (tag, sub)
        assert os.path.exists(os.path.join(prof, "round3_%s_summary.txt" % tag))


def test_c_abi_pointers_reject_strided_views():
    """The library knows no strides: a transposed or column-sliced tensor handed to the ctypes layer must fail loudly
    instead of being read as a dense row-major buffer (a replay-noise tensor built from transposed views once was)."""
    import torch
    from vaenmf.engine import _ptr
    t = torch.zeros(6, 4)
    assert _ptr(None) is None and _ptr(t) is not None and _ptr(t[:3]) is not None
    with pytest.raises(ValueError):
        _ptr(t.t())
    with pytest.raises(ValueError):
        _ptr(t[:, :2])
    assert _ptr(t[:, :2], rows_strided=True) is not None          # vaenmf_dense takes a row stride
    with pytest.raises(ValueError):
        _ptr(t.t(), rows_strided=True)

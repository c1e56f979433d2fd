#!/usr/bin/env python3
"""Crops of the reference's committed processed utterances (build container only; data, no code).

/root/reference/data/subset/processed/CSR-1-WSJ-0/WAV/wsj0/{si_tr_s/011, si_dt_05/050, si_et_05/440} holds nine utterances x
{_s, _n, _x}.wav (RIFF PCM-16 mono 16 kHz, written by the reference's scripts/create_test_set.py).  The driver / metrics
tests (rows f1 / f2) need real files of DIFFERENT lengths; the full set is 6.6 MB, so crops are committed:

  tests/golden/processed_subset.npz    si_et_05/440/440c020{a,b,c}, samples 8000 .. 27999 of each (equal lengths; round 2)
  tests/golden/processed_subset9.npz   all nine utterances, ragged: samples 8000 .. 8000+T_u with nine different T_u between
                                       1.0 and 1.95 s; keys <u>_{s,n,x} int16, rel (paths relative to the dataset root, in
                                       speech_list order per subset), subset ('train' / 'validation' / 'test'), fs.

Usage: python tests/golden/make_processed_subset.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PROC = "/root/reference/data/subset/processed/"
OFF = 8000
LEN9 = [20000, 27111, 16384, 24000, 31000, 18000, 22222, 29000, 25600]


def read_wav_int16(path):
    """Minimal RIFF/WAVE PCM16 mono reader (soundfile is not installed)."""
    b = open(path, "rb").read()
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE"
    pos, fs = 12, None
    while pos < len(b):
        cid, sz = b[pos:pos + 4], int.from_bytes(b[pos + 4:pos + 8], "little")
        if cid == b"fmt ":
            assert int.from_bytes(b[pos + 8:pos + 10], "little") == 1 and int.from_bytes(b[pos + 10:pos + 12], "little") == 1
            fs = int.from_bytes(b[pos + 12:pos + 16], "little")
            assert int.from_bytes(b[pos + 22:pos + 24], "little") == 16
        if cid == b"data":
            return np.frombuffer(b[pos + 8:pos + 8 + sz], dtype="<i2").copy(), fs
        pos += 8 + sz + (sz & 1)
    raise ValueError("no data chunk")


def main():
    sets = [("train", "CSR-1-WSJ-0/WAV/wsj0/si_tr_s/011/", ["011a010a", "011a010b", "011a010c"]),
            ("validation", "CSR-1-WSJ-0/WAV/wsj0/si_dt_05/050/", ["050a050a", "050a050b", "050a050c"]),
            ("test", "CSR-1-WSJ-0/WAV/wsj0/si_et_05/440/", ["440c020a", "440c020b", "440c020c"])]
    out9, rel9, sub9 = {}, [], []
    i = 0
    for subset, d, names in sets:
        for nm in names:
            T = LEN9[i]
            for k in "snx":
                w, fs = read_wav_int16(PROC + d + nm + "_%s.wav" % k)
                assert fs == 16000 and len(w) >= OFF + T
                out9["u%d_%s" % (i, k)] = w[OFF:OFF + T]
            rel9.append(d + nm + ".wav")
            sub9.append(subset)
            i += 1
    np.savez_compressed(os.path.join(HERE, "processed_subset9.npz"), rel=np.array(rel9), subset=np.array(sub9), fs=16000, **out9)
    # the round-2 fixture, reproduced (and checked against the committed file)
    out3 = {}
    for u, nm in zip("abc", sets[2][2]):
        for k in "snx":
            out3["%s_%s" % (u, k)] = read_wav_int16(PROC + sets[2][1] + nm + "_%s.wav" % k)[0][OFF:OFF + 20000]
    old = os.path.join(HERE, "processed_subset.npz")
    if os.path.exists(old):
        z = np.load(old)
        assert all(np.array_equal(z[k], v) for k, v in out3.items()), "processed_subset.npz differs from the reference's files"
        print("processed_subset.npz reproduced bit for bit")
    else:
        np.savez_compressed(old, rel=np.array([sets[2][1] + nm + ".wav" for nm in sets[2][2]]), fs=16000, **out3)
    print("processed_subset9.npz:", [(r.split("/")[-1], t) for r, t in zip(rel9, LEN9)], os.path.getsize(os.path.join(HERE, "processed_subset9.npz")), "bytes")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Known answers the reference holds as pickles, extracted WITHOUT unpickling (build container only).

The reference's only golden STFT output is data/subset/pickle/CSR-1-WSJ-0/si_et_05_frames.p: |STFT|^2 of its three
raw si_et_05/440 utterances, written by tests/dataset/test_csr1_wjs0_dataset.py:17-83 (drop the first 0.1 s,
divide by the peak, stft(wlen 64 ms, hann, hop 25 %), |.|^2, concatenated along the frame axis).  Its per-utterance
input SNRs for run_metrics_M1.py:149-151 are data/subset/processed/.../si_et_05_snr_db.p.  Both are pickles; no
pickle loader is run on them.  `read_plain_pickle` below walks the opcode stream with pickletools.genops -- which
only DECODES opcodes, it constructs nothing and imports nothing -- and rebuilds exactly three kinds of value by
hand: numpy ndarrays (from the shape / dtype-string / order flag / raw bytes operands of numpy's _reconstruct
state), Python lists and floats.  Any other global, reduce or opcode raises.

Output (data only): tests/golden/stft_frames.npz -- int16 samples of utterance 440c020a (from its NIST SPHERE file:
1024-byte ASCII header + little-endian PCM), the first and last 96 frames of its |STFT|^2 from the pickle, the
frame counts of the three utterances; tests/golden/stft_frames_tr_dt.npz -- the same for the reference's two other golden
STFT outputs, si_tr_s_frames.p (513, 972) and si_dt_05_frames.p (513, 976): PCM of the first two utterances of each set and
frames / per-frame sums of their |STFT|^2; tests/golden/snr_db.npz -- the input-SNR lists.
Usage: python tests/golden/extract_ref_pickles.py
"""
import os
import pickletools
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/data/subset"


class _Global:
    def __init__(self, module, name):
        self.module, self.name = module, name


class _NDArrayStub:
    pass


class _DTypeStub:
    def __init__(self, code):
        self.code, self.order = code, "|"


_MARK = object()


def read_plain_pickle(path):
    """ndarray / list / float / int / str / tuple / None / bool values only; nothing from the file is executed."""
    stack, memo = [], []

    def pop_to_mark():
        out = []
        while True:
            v = stack.pop()
            if v is _MARK:
                return out[::-1]
            out.append(v)

    with open(path, "rb") as f:
        for op, arg, pos in pickletools.genops(f):
            n = op.name
            if n in ("PROTO", "FRAME"):
                pass
            elif n in ("SHORT_BINUNICODE", "BINUNICODE", "SHORT_BINBYTES", "BINBYTES", "BINBYTES8", "BININT", "BININT1", "BININT2", "BINFLOAT", "LONG1"):
                stack.append(arg)
            elif n == "NONE":
                stack.append(None)
            elif n == "NEWTRUE":
                stack.append(True)
            elif n == "NEWFALSE":
                stack.append(False)
            elif n == "MEMOIZE":
                memo.append(stack[-1])
            elif n in ("BINGET", "LONG_BINGET"):
                stack.append(memo[arg])
            elif n == "MARK":
                stack.append(_MARK)
            elif n == "TUPLE1":
                stack[-1:] = [(stack[-1],)]
            elif n == "TUPLE2":
                stack[-2:] = [tuple(stack[-2:])]
            elif n == "TUPLE3":
                stack[-3:] = [tuple(stack[-3:])]
            elif n == "TUPLE":
                stack.append(tuple(pop_to_mark()))
            elif n == "EMPTY_TUPLE":
                stack.append(())
            elif n == "EMPTY_LIST":
                stack.append([])
            elif n == "APPEND":
                v = stack.pop()
                stack[-1].append(v)
            elif n == "APPENDS":
                items = pop_to_mark()
                stack[-1].extend(items)
            elif n == "STACK_GLOBAL":
                name, module = stack.pop(), stack.pop()
                stack.append(_Global(module, name))
            elif n == "REDUCE":
                args, fn = stack.pop(), stack.pop()
                if not isinstance(fn, _Global):
                    raise ValueError("REDUCE of a non-global at %d" % pos)
                key = (fn.module, fn.name)
                if key in (("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct")):
                    stack.append(_NDArrayStub())
                elif key == ("numpy", "dtype"):
                    stack.append(_DTypeStub(args[0]))
                elif key in (("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar")):
                    dt, raw = args
                    stack.append(np.frombuffer(raw, dtype=np.dtype(dt.order.replace("|", "") + dt.code if dt.order in "<>" else dt.code))[0].item())
                else:
                    raise ValueError("unexpected callable %s.%s in %s" % (fn.module, fn.name, path))
            elif n == "BUILD":
                state, obj = stack.pop(), stack[-1]
                if isinstance(obj, _DTypeStub):
                    obj.order = state[1]                                   # (version, endian, ...)
                elif isinstance(obj, _NDArrayStub):
                    ver, shape, dt, fortran, raw = state
                    code = (dt.order if dt.order in "<>" else "") + dt.code
                    a = np.frombuffer(raw, dtype=np.dtype(code))
                    stack[-1] = a.reshape(shape[::-1]).T if fortran else a.reshape(shape)
                else:
                    raise ValueError("BUILD on %r" % type(obj))
            elif n == "STOP":
                break
            else:
                raise ValueError("opcode %s not allowed (%s at %d)" % (n, path, pos))
    assert len(stack) == 1
    return stack[0]


def read_sphere_pcm16(path):
    """NIST SPHERE, uncompressed: 1024-byte ASCII header, little-endian 16-bit PCM."""
    raw = open(path, "rb").read()
    head = raw[:1024].decode("ascii", "replace")
    assert head.startswith("NIST_1A") and "sample_byte_format -s2 01" in head and "sample_coding -s3 pcm" in head, head[:200]
    return np.frombuffer(raw[1024:], dtype="<i2").copy()


def main():
    frames = read_plain_pickle(os.path.join(REF, "pickle/CSR-1-WSJ-0/si_et_05_frames.p"))
    assert frames.shape == (513, 2009) and frames.dtype == np.float32, (frames.shape, frames.dtype)
    wavs = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(REF, "raw/CSR-1-WSJ-0/WAV/wsj0/si_et_05")) for f in fs if f.endswith(".wav"))
    pcm = [read_sphere_pcm16(w) for w in wavs]
    fs, nfft, hop = 16000, 1024, 256
    counts = []
    for x in pcm:                                      # frame count per utterance: stft.py:48-53 end-pad rule, librosa centre framing
        T = len(x) - int(0.1 * fs)
        pad = hop if int(np.ceil(T / fs / 64e-3 / 0.25)) != int(T / fs / 64e-3 / 0.25) else 0
        counts.append(1 + (T + pad) // hop)
    assert sum(counts) == frames.shape[1], (counts, frames.shape)
    n0 = counts[0]
    np.savez_compressed(os.path.join(HERE, "stft_frames.npz"), pcm_a=pcm[0], name_a=os.path.basename(wavs[0]), frame_counts=np.array(counts),
                        head=np.ascontiguousarray(frames[:, :96]), tail=np.ascontiguousarray(frames[:, n0 - 96:n0]),
                        col_sums=frames[:, :n0].sum(0, dtype=np.float64), peak_b=np.abs(pcm[1][1600:]).max(), peak_c=np.abs(pcm[2][1600:]).max(),
                        first_b=np.ascontiguousarray(frames[:, n0:n0 + 4]))
    # ---- the two other golden STFT outputs of the same reference test (train: si_tr_s/011, validation: si_dt_05/050):
    # (513, 972) and (513, 976).  Per set: PCM of its first utterance, the first / last 48 frames and the per-frame sums of
    # that utterance's |STFT|^2, the frame counts, and the first 4 frames + the peak of the second utterance.
    extra = {}
    for tag, sub, shape in (("tr", "si_tr_s", (513, 972)), ("dt", "si_dt_05", (513, 976))):
        fr = read_plain_pickle(os.path.join(REF, "pickle/CSR-1-WSJ-0/%s_frames.p" % sub))
        assert fr.shape == shape and fr.dtype == np.float32, (fr.shape, fr.dtype)
        ws = sorted(os.path.join(dp, f) for dp, _, fs2 in os.walk(os.path.join(REF, "raw/CSR-1-WSJ-0/WAV/wsj0/" + sub)) for f in fs2 if f.endswith(".wav"))
        pc = [read_sphere_pcm16(w) for w in ws]
        cn = []
        for x in pc:
            T = len(x) - int(0.1 * fs)
            pad = hop if int(np.ceil(T / fs / 64e-3 / 0.25)) != int(T / fs / 64e-3 / 0.25) else 0
            cn.append(1 + (T + pad) // hop)
        assert sum(cn) == fr.shape[1], (cn, fr.shape)
        m0 = cn[0]
        extra.update({tag + "_pcm_a": pc[0], tag + "_name_a": os.path.basename(ws[0]), tag + "_frame_counts": np.array(cn),
                      tag + "_head": np.ascontiguousarray(fr[:, :48]), tag + "_tail": np.ascontiguousarray(fr[:, m0 - 48:m0]),
                      tag + "_col_sums": fr[:, :m0].sum(0, dtype=np.float64), tag + "_pcm_b": pc[1],
                      tag + "_first_b": np.ascontiguousarray(fr[:, m0:m0 + 4]), tag + "_col_sums_b": fr[:, m0:m0 + cn[1]].sum(0, dtype=np.float64)})
        print(sub, "frames", fr.shape, "counts", cn, [os.path.basename(w) for w in ws])
    np.savez_compressed(os.path.join(HERE, "stft_frames_tr_dt.npz"), **extra)
    out = {}
    for dp, _, fs_ in os.walk(REF):
        for f in fs_:
            if f.endswith("_snr_db.p"):
                v = read_plain_pickle(os.path.join(dp, f))
                out[os.path.relpath(os.path.join(dp, f), REF).replace("/", "__")[:-2]] = np.asarray(v, dtype=np.float64)
    np.savez(os.path.join(HERE, "snr_db.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, v[:8])
    print("frames", frames.shape, "counts", counts, "wavs", [os.path.basename(w) for w in wavs])


if __name__ == "__main__":
    main()

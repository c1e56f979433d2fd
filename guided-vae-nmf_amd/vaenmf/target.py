"""Label / guide front-ends of the M2 path on the device: the build's counterpart of
python/processing/target.py (clean_speech_IBM :7-28, clean_speech_VAD :30-50, the noise-robust
variants :52-102, ideal_wiener_mask :104-116).  Same names, arguments, shapes and dtypes; the
work runs in the HIP library (csrc/labels.hip) and there is no CPU fallback.

`*_batch` variants take the frame-major device spectrogram of a whole batch ([NT][Fs], as
BatchEngine / stft_batch hold it) and return device labels ready for `set_labels`."""
import numpy as np
import torch

from . import _lib
from ._lib import lib, check


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _soft_values(quantile_weight):
    """0.5 + w (mask - 0.5), np.round (half to even), float32 -- target.py:24-27."""
    return float(np.float32(np.round(0.5 - 0.5 * quantile_weight))), float(np.float32(np.round(0.5 + 0.5 * quantile_weight)))


def lorenz_labels_batch(X, frame_counts, F, mode, quantile_fraction=0.98, quantile_weight=0.999, want_thresholds=False):
    """X device complex64 [NT][Fs] (or its float32 view [NT][Fs][2], as stft_batch returns it); returns device
    float32 [NT][F] (mode 'ibm') or [NT] (mode 'vad')."""
    if not X.is_cuda:
        raise RuntimeError("lorenz_labels_batch needs the spectrogram on the GPU (no CPU fallback)")
    if not X.is_complex():
        X = torch.view_as_complex(X.contiguous())
    NT, Fs = X.shape
    off = np.concatenate([[0], np.cumsum(frame_counts)]).astype(np.int32)
    assert off[-1] == NT
    m = {"ibm": _lib.LABEL_IBM, "vad": _lib.LABEL_VAD}[mode]
    U = len(frame_counts)
    nbytes = lib().vaenmf_lorenz_work_bytes(NT, F, U, m)
    work = torch.empty(int(nbytes), dtype=torch.uint8, device=X.device)
    out = torch.empty((NT, F) if m == _lib.LABEL_IBM else (NT,), dtype=torch.float32, device=X.device)
    thr = torch.empty(U, dtype=torch.float32, device=X.device)
    lo, hi = _soft_values(quantile_weight)
    Xr = torch.view_as_real(X.contiguous())
    check(lib().vaenmf_lorenz_labels(_ptr(Xr), U, off.ctypes.data, F, Fs, m, float(quantile_fraction), lo, hi, _ptr(out), F,
                                     _ptr(thr), _ptr(work), int(nbytes), _stream()))
    return (out, thr) if want_thresholds else out


def _to_frames(observations, device):
    obs = np.asarray(observations)
    if obs.dtype != np.complex64:
        raise TypeError("this build computes the labels in the complex64 / float32 arithmetic of the reference's "
                        "STFT (dtype='complex64'); got %s" % obs.dtype)
    return torch.from_numpy(np.ascontiguousarray(obs.T)).to(device)       # [N][F]


def clean_speech_IBM(observations, quantile_fraction=0.98, quantile_weight=0.999, device="cuda:0"):
    """observations complex64 (F, N) -> float32 (F, N) of 0/1 (target.py:7-28)."""
    X = _to_frames(observations, device)
    y = lorenz_labels_batch(X, [X.shape[0]], X.shape[1], "ibm", quantile_fraction, quantile_weight)
    return np.ascontiguousarray(y.cpu().numpy().T)


def clean_speech_VAD(observations, quantile_fraction=0.98, quantile_weight=0.999, device="cuda:0"):
    """-> float32 (1, N) (target.py:30-50)."""
    X = _to_frames(observations, device)
    y = lorenz_labels_batch(X, [X.shape[0]], X.shape[1], "vad", quantile_fraction, quantile_weight)
    return y.cpu().numpy()[None]


def noise_robust_clean_speech_VAD(observations, quantile_fraction_begin=0.93, quantile_fraction_end=0.99,
                                  quantile_weight=0.999, device="cuda:0"):
    """target.py:52-76: active from the first frame of the strict VAD to (excluding) the last frame of the lenient one."""
    vad = clean_speech_VAD(observations, quantile_fraction_begin, quantile_weight, device)[0]
    end = clean_speech_VAD(observations, quantile_fraction_end, quantile_weight, device)[0]
    b, e = np.nonzero(vad)[0][0], np.nonzero(end)[0][-1]
    vad[b:e] = 1
    return vad[None]


def noise_robust_clean_speech_IBM(observations, vad_quantile_fraction_begin=0.93, vad_quantile_fraction_end=0.99,
                                  ibm_quantile_fraction=0.999, quantile_weight=0.999, device="cuda:0"):
    """target.py:78-102."""
    vad = noise_robust_clean_speech_VAD(observations, vad_quantile_fraction_begin, vad_quantile_fraction_end, quantile_weight, device)
    return clean_speech_IBM(observations, ibm_quantile_fraction, quantile_weight, device) * vad


def ideal_wiener_mask(speech_tf, noise_tf, eps=1e-8, device="cuda:0"):
    """|S|^2 / (|S|^2 + |N|^2 + eps), float32, same shape (target.py:104-116)."""
    s = np.asarray(speech_tf)
    n = np.asarray(noise_tf)
    if s.dtype != np.complex64 or n.dtype != np.complex64 or s.shape != n.shape:
        raise TypeError("ideal_wiener_mask: two complex64 arrays of one shape expected")
    S = torch.view_as_real(torch.from_numpy(np.ascontiguousarray(s)).to(device))
    N = torch.view_as_real(torch.from_numpy(np.ascontiguousarray(n)).to(device))
    out = torch.empty(s.shape, dtype=torch.float32, device=device)
    check(lib().vaenmf_wiener_mask(_ptr(S), _ptr(N), int(s.size), float(eps), _ptr(out), _stream()))
    return out.cpu().numpy()

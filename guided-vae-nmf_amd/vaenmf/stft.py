"""stft / istft with the reference's signatures (python/processing/stft.py:16-24,
66-73) running on the GPU (vaenmf_stft_batch / vaenmf_istft_batch), plus batched
device-resident variants used by the pipeline."""
import ctypes as C

import numpy as np
import torch

from ._lib import check, lib
from .engine import _ptr, _stream


def frame_geometry(n_samples, fs, wlen_sec, hop_percent):
    """(nfft, hop, n_frames, padded_len); raises ValueError like stft.py:37-38."""
    nfft, hop, nfr, npad = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    rc = lib().vaenmf_stft_num_frames(int(n_samples), float(fs), float(wlen_sec), float(hop_percent),
                                      C.byref(nfft), C.byref(hop), C.byref(nfr), C.byref(npad))
    if rc != 0:
        msg = lib().vaenmf_last_error().decode()
        if "not an integer" in msg:
            raise ValueError(msg)
        raise NotImplementedError(msg)
    return nfft.value, hop.value, nfr.value, npad.value


_TABLES = {}


def _remember(key, tab, limit=16):
    if len(_TABLES) >= limit:
        _TABLES.pop(next(iter(_TABLES)))
    _TABLES[key] = tab


def stft_batch(wav, sample_counts, fs, wlen_sec, hop_percent, Fs=None, device="cuda:0"):
    """wav: device float32 [sum T] (utterances concatenated).  Returns (X [NT,Fs,2], frame_counts)."""
    dev = torch.device(device)
    key = ("stft", tuple(int(t) for t in sample_counts), fs, wlen_sec, hop_percent, str(dev))
    tab = _TABLES.get(key)
    if tab is None:       # index tables on the device, cached per batch shape (pageable uploads make the host wait for the GPU)
        geo = [frame_geometry(t, fs, wlen_sec, hop_percent) for t in sample_counts]
        fc_ = [g[2] for g in geo]
        tab = (geo[0][0], geo[0][1], fc_,
               torch.tensor(np.concatenate([[0], np.cumsum(sample_counts)]), dtype=torch.int64, device=dev),
               torch.tensor(np.concatenate([[0], np.cumsum(fc_)]), dtype=torch.int32, device=dev),
               torch.repeat_interleave(torch.arange(len(fc_), dtype=torch.int32), torch.tensor(fc_)).to(dev),
               torch.tensor([g[3] for g in geo], dtype=torch.int32, device=dev))
        _remember(key, tab)
    nfft, hop, fc, soff, foff, futt, plen = tab
    fc = list(fc)
    F = nfft // 2 + 1
    Fs = Fs or (F + 15) // 16 * 16
    NT = int(sum(fc))
    X = torch.empty(NT, Fs, 2, device=dev, dtype=torch.float32)
    check(lib().vaenmf_stft_batch(_ptr(wav), NT, _ptr(soff), _ptr(foff), _ptr(futt), _ptr(plen), nfft, hop, Fs, _ptr(X), _stream()))
    return X, fc


def istft_batch(S, frame_counts, sample_counts, nfft, hop, device="cuda:0"):
    """S: device [NT,Fs,2] complex64 -> device float32 [sum T] (max_len = sample_counts[u])."""
    dev = torch.device(device)
    key = ("istft", tuple(int(t) for t in sample_counts), tuple(int(t) for t in frame_counts), str(dev))
    tab = _TABLES.get(key)
    if tab is None:
        tab = (torch.tensor(np.concatenate([[0], np.cumsum(sample_counts)]), dtype=torch.int64, device=dev),
               torch.tensor(np.concatenate([[0], np.cumsum(frame_counts)]), dtype=torch.int32, device=dev))
        _remember(key, tab)
    soff, foff = tab
    NT, Fs = S.shape[0], S.shape[1]
    work = torch.empty(NT, nfft, device=dev, dtype=torch.float32)
    out = torch.empty(int(sum(sample_counts)), device=dev, dtype=torch.float32)
    check(lib().vaenmf_istft_batch(_ptr(S), len(frame_counts), NT, _ptr(soff), _ptr(foff), nfft, hop, Fs, _ptr(work), _ptr(out), _stream()))
    return out


def stft(x, fs=16e3, wlen_sec=50e-3, win="hann", hop_percent=0.25, center=True, pad_mode="reflect",
         pad_at_end=True, dtype="complex64"):
    """Reference signature (stft.py:16-24).  Returns numpy complex64 (F, n_frames)."""
    if win != "hann" or not center or pad_mode != "reflect" or not pad_at_end:
        raise NotImplementedError("only the reference defaults (hann, center, reflect, pad_at_end) are built")
    x = np.asarray(x)
    wav = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    X, fc = stft_batch(wav, [len(x)], fs, wlen_sec, hop_percent)
    nfft = frame_geometry(len(x), fs, wlen_sec, hop_percent)[0]
    F = nfft // 2 + 1
    Xc = np.ascontiguousarray(X[:, :F].cpu().numpy()).view(np.complex64).reshape(fc[0], F)
    return Xc.T.astype(dtype)


def istft(Sxx, fs=16000, wlen_sec=50e-3, win="hann", hop_percent=0.25, center=True, dtype="float32", max_len=None):
    """Reference signature (stft.py:66-73).  Sxx numpy complex (F, n_frames)."""
    if wlen_sec * fs != int(wlen_sec * fs):
        raise ValueError("wlen_sample of iSTFT is not an integer.")
    nfft = int(wlen_sec * fs)
    hop = int(hop_percent * nfft)
    F, nfr = Sxx.shape
    Fs = (F + 15) // 16 * 16
    S = np.zeros((nfr, Fs), np.complex64)
    S[:, :F] = np.asarray(Sxx).T
    Sd = torch.from_numpy(S.view(np.float32).reshape(nfr, Fs, 2)).cuda()
    T = int(max_len) if max_len else hop * (nfr - 1)
    out = istft_batch(Sd, [nfr], [T], nfft, hop)
    return out.cpu().numpy().astype(dtype)

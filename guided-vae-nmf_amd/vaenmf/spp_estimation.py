"""SPP-based noise-PSD / speech-presence estimator on the device: the build's counterpart of
python/models/spp_estimation.py (timo_mask_estimation :163-183, timo_vad_estimation :185-214,
timo_noise_estimation :218-235; the estimator itself :17-160).  Same names and shapes; the
recursion runs in the HIP library (csrc/labels.hip: one thread per utterance and bin, float64
state like the reference's numpy arrays).  No CPU fallback."""
import numpy as np
import torch

from ._lib import lib, check

# spp_estimation.py:10-14
SPP_FIX_SMOOTH = 0.8
SPP_PROB_SMOOTH = 0.9
SPP_PRIOR = 0.5
SPP_SNR_OPT_DB = 15
SPP_NUM_FRAMES_INIT = 10


def _stream():
    return torch.cuda.current_stream().cuda_stream


def spp_batch(per, frame_counts, F, want_psd=False, fixed_smooth=SPP_FIX_SMOOTH, prob_smooth=SPP_PROB_SMOOTH, prior=SPP_PRIOR,
              snr_opt_db=SPP_SNR_OPT_DB, num_frames_init=SPP_NUM_FRAMES_INIT):
    """per device float32 [NT][ld] periodograms of a batch of utterances -> device float32 spp [NT][F] (and the
    noise PSD [NT][F] when want_psd)."""
    if not per.is_cuda:
        raise RuntimeError("spp_batch needs the periodogram on the GPU (no CPU fallback)")
    per = per.contiguous()
    NT, ld = per.shape
    off = torch.tensor(np.concatenate([[0], np.cumsum(frame_counts)]), dtype=torch.int32, device=per.device)
    spp = torch.empty(NT, F, dtype=torch.float32, device=per.device)
    psd = torch.empty(NT, F, dtype=torch.float32, device=per.device) if want_psd else None
    check(lib().vaenmf_spp_estimate(per.data_ptr(), ld, len(frame_counts), off.data_ptr(), F, float(fixed_smooth), float(prob_smooth),
                                    float(prior), float(snr_opt_db), int(num_frames_init), spp.data_ptr(),
                                    None if psd is None else psd.data_ptr(), F, _stream()))
    return (spp, psd) if want_psd else spp


def timo_mask_estimation(spectrogram, device="cuda:0"):
    """spectrogram: power spectrogram |Y|^2 (freq_bins, frames) -> SPP mask of the same shape and dtype (:163-183)."""
    sp = np.asarray(spectrogram)
    per = torch.from_numpy(np.ascontiguousarray(sp.T, dtype=np.float32)).to(device)
    spp = spp_batch(per, [per.shape[0]], per.shape[1])
    return np.ascontiguousarray(spp.cpu().numpy().T).astype(sp.dtype)


def timo_vad_estimation(spectrogram, device="cuda:0"):
    """Frame-level SPP of the power summed over the bins (:185-214) -> (frames,)."""
    sp = np.asarray(spectrogram)
    s = sp.sum(axis=0)
    per = torch.from_numpy(np.ascontiguousarray(s[:, None], dtype=np.float32)).to(device)
    spp = spp_batch(per, [per.shape[0]], 1)
    return spp.cpu().numpy()[:, 0].astype(s.dtype)


def timo_noise_estimation(spectrogram, mask, device="cuda:0"):
    """Noise PSD from a given SPP mask (:218-235; the reference's v_spp_in branch never updates its old PSD, so this
    is (1 - fixed_smooth) (1 - mask) |Y|^2)."""
    sp = np.asarray(spectrogram)
    per = torch.from_numpy(np.ascontiguousarray(sp, dtype=np.float32)).to(device)
    m = torch.from_numpy(np.ascontiguousarray(np.asarray(mask), dtype=np.float32)).to(device)
    out = torch.empty_like(per)
    check(lib().vaenmf_spp_noise_given(per.data_ptr(), m.data_ptr(), per.numel(), SPP_FIX_SMOOTH, out.data_ptr(), _stream()))
    return out.cpu().numpy().astype(sp.dtype)

"""BatchEngine: host-side owner of the device state of a batch of utterances and
thin caller of the C ABI (include/vaenmf.h).  PyTorch-ROCm is used only for device
memory and streams; every numerical step of the hot path runs in libvaenmf.so."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, lib

LAT, HID = 32, 128


def _ptr(t, rows_strided=False):
    """Raw address of a tensor handed to the C ABI (row-major, dense: the library knows no strides; `rows_strided`: a row
    stride goes along, as for vaenmf_dense's input).  A transposed / sliced view would be read as something else entirely."""
    if t is None:
        return None
    if not (t.is_contiguous() or (rows_strided and t.dim() == 2 and t.stride(1) == 1)):
        raise ValueError("the C ABI takes dense row-major buffers; got strides %s for shape %s (call .contiguous())" % (tuple(t.stride()), tuple(t.shape)))
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _np32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def decoder_params_from_state(sd, prefix="decoder."):
    """Pick the decoder weights out of a state_dict in the reference key layout
    (decoder.hidden.{i}.{weight,bias}, decoder.reconstruction.{weight,bias})."""
    get = lambda k: _np32(sd[prefix + k].detach().cpu().numpy() if hasattr(sd[prefix + k], "detach") else sd[prefix + k])
    n = 0
    while (prefix + "hidden.%d.weight" % n) in sd:
        n += 1
    if n == 1:            # h_dim = [128] (scripts/evaluate_M1.py:44-85 lists such checkpoints): [W1, b1, W3, b3]
        return [get("hidden.0.weight"), get("hidden.0.bias"), get("reconstruction.weight"), get("reconstruction.bias")]
    if n != 2:
        raise NotImplementedError("this build runs decoders with 1 or 2 hidden layers of 128 units (got %d layers)" % n)
    return [get("hidden.0.weight"), get("hidden.0.bias"), get("hidden.1.weight"), get("hidden.1.bias"),
            get("reconstruction.weight"), get("reconstruction.bias")]


def classifier_layers_from_state(sd, two_classes=False, bn_eps=1e-5):
    """[(W, b) hidden..., (W, b) output] for BatchEngine.classify / pipeline.MaskEnhancer from a Classifier state_dict
    (models.py:41-62: `hidden.{i}.*`, `output_layer.*`).  With batch_norm=True the ModuleList alternates Linear and
    BatchNorm1d and the reference applies relu to EVERY module (relu(BN(relu(Linear))), models.py:50-52, 59-60): in eval mode
    (scripts/reconstruct_dnn_classif.py:129) a BatchNorm1d is the per-feature affine s = gamma / sqrt(running_var + eps),
    t = beta - running_mean * s, handed to the dense kernel as one more ReLU layer with a diagonal weight.
    two_classes: Classifier2Classes (models.py:64-88): softmax over the two classes of output_layer(x).view(-1, 2, y_dim);
    the returned output layer is (W[:y] - W[y:], b[:y] - b[y:]), whose sigmoid is the class-0 probability (class 1 = 1 - it)."""
    get = lambda k: _np32(sd[k].detach().cpu().numpy() if hasattr(sd[k], "detach") else sd[k])
    layers, i = [], 0
    while "hidden.%d.weight" % i in sd:
        w = get("hidden.%d.weight" % i)
        if w.ndim == 2:
            layers.append((w, get("hidden.%d.bias" % i)))
        else:
            s_ = w.astype(np.float64) / np.sqrt(get("hidden.%d.running_var" % i).astype(np.float64) + bn_eps)
            t_ = get("hidden.%d.bias" % i).astype(np.float64) - get("hidden.%d.running_mean" % i).astype(np.float64) * s_
            layers.append((_np32(np.diag(s_)), _np32(t_)))
        i += 1
    wo, bo = get("output_layer.weight"), get("output_layer.bias")
    if two_classes:
        y = wo.shape[0] // 2
        wo, bo = _np32(wo[:y].astype(np.float64) - wo[y:].astype(np.float64)), _np32(bo[:y].astype(np.float64) - bo[y:].astype(np.float64))
    layers.append((wo, bo))
    return layers


def latent_dim_from_state(sd):
    """z_dim of a model from its state_dict (encoder.sample.mu.weight is (z_dim, h))."""
    return int(sd["encoder.sample.mu.weight"].shape[0])


class BatchEngine:
    def __init__(self, F, K, decoder, precision="bf16x3", device="cuda:0", max_frames=1 << 16, max_utts=256, z_dim=LAT):
        """decoder = [W1 (H,L+Dy), b1, W2 (H,H), b2, W3 (F,H), b3] float32 numpy (nn.Linear layout), or [W1, b1, W3, b3]
        for a decoder with one hidden layer (h_dim = [128]).  z_dim = L: 32 or 16 (16 runs on the 32-wide first layer with
        zero padding: Z / Zs keep 32 columns, the last 16 stay zero)."""
        if not torch.cuda.is_available():
            raise RuntimeError("vaenmf needs a ROCm GPU (MI355X); there is no CPU path")
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        dec = [_np32(a) for a in decoder]
        if len(dec) == 4:
            W1, b1, W3, b3 = dec
            W2 = b2 = None
        else:
            W1, b1, W2, b2, W3, b3 = dec
        if W1.shape[0] != HID or (W2 is not None and W2.shape != (HID, HID)) or W3.shape != (F, HID):
            raise NotImplementedError("this build runs decoders z(32|16)+y -> %d [-> %d] -> F; got %s %s %s"
                                      % (HID, HID, W1.shape, None if W2 is None else W2.shape, W3.shape))
        if int(z_dim) not in (16, LAT):
            raise NotImplementedError("latent dim %d: this build supports 16 and %d (z_dim 128 would need 4 k-steps in the "
                                      "first layer and 128 latents per frame in registers)" % (z_dim, LAT))
        self.F, self.K, self.L = int(F), int(K), int(z_dim)
        self.Dy = W1.shape[1] - self.L
        if self.Dy < 0:
            raise NotImplementedError("first decoder layer has %d inputs, fewer than the latent dim %d" % (W1.shape[1], self.L))
        self.precision = {"bf16x3": _lib.PREC_BF16X3, "bf16": _lib.PREC_BF16}[precision]
        self._max_frames, self._max_utts = int(max_frames), int(max_utts)
        cfg = _lib.Config(self.F, self.K, self.L, HID, HID if W2 is not None else 0, int(max_frames), int(max_utts), self.precision)
        self._plan = C.c_void_p()
        check(lib().vaenmf_plan_create(C.byref(cfg), C.byref(self._plan)))
        check(lib().vaenmf_set_decoder_weights(self._plan, W1.ctypes.data, W1.shape[1], b1.ctypes.data, None if W2 is None else W2.ctypes.data,
                                               None if b2 is None else b2.ctypes.data, W3.ctypes.data, b3.ctypes.data))
        self.Fs = lib().vaenmf_plan_query(self._plan, _lib.Q_FS)
        self.Kp = lib().vaenmf_plan_query(self._plan, _lib.Q_KP)
        self.NT = 0
        self.B1 = None

    def close(self):
        if getattr(self, "_plan", None):
            lib().vaenmf_plan_destroy(self._plan)
            self._plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ batch
    def bind(self, frame_counts, Rcap, seeds=None):
        fc = [int(n) for n in frame_counts]
        self.frame_off = np.concatenate([[0], np.cumsum(fc)]).astype(np.int32)
        self.U, self.NT, self.Rcap = len(fc), int(self.frame_off[-1]), int(Rcap)
        sd = None
        if seeds is not None:
            sd = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64))
        check(lib().vaenmf_bind_batch_async(self._plan, self.U, self.frame_off.ctypes.data, None if sd is None else sd.ctypes.data, _stream()))
        dev, f32 = self.device, torch.float32
        NT, Fs, Kp = self.NT, self.Fs, self.Kp
        # Device buffers are allocated once at the plan's capacity (max_frames, max_utts) and every bind takes leading
        # views: a fresh 300 MB Zs per batch cost ~9 ms of allocator time per 64-utterance batch, and the 1000-utterance
        # job alternates batches of 63 and 62 utterances.
        if getattr(self, "_cap_key", None) != self.Rcap:
            MF, MU = self._max_frames, self._max_utts
            self._bX = torch.empty(MF, Fs, 2, device=dev, dtype=f32)      # complex64, interleaved
            self._bX2 = torch.empty(MF, Fs, device=dev, dtype=f32)
            self._bW = torch.empty(MU, Fs, Kp, device=dev, dtype=f32)
            self._bHt = torch.empty(MF, Kp, device=dev, dtype=f32)
            self._bg = torch.empty(MF, device=dev, dtype=f32)
            self._bZ = torch.empty(MF, LAT, device=dev, dtype=f32)
            self._bZs = torch.empty(MF, self.Rcap, LAT, device=dev, dtype=f32)
            self._bcost = torch.empty(MF, device=dev, dtype=torch.float64)
            self._bS = torch.empty(MF, Fs, 2, device=dev, dtype=f32)      # outputs of run(): fixed addresses, so that a repeated
            self._bN = torch.empty(MF, Fs, 2, device=dev, dtype=f32)      # call has the signature vaenmf_em_run replays as a graph
            self._bcu = {}                                                # niter -> [max_utts, niter] float64
            self._cap_key = self.Rcap
        self.X, self.X2, self.W, self.Ht = self._bX[:NT], self._bX2[:NT], self._bW[:self.U], self._bHt[:NT]
        self.g, self.Z, self.Zs, self.cost_frames = self._bg[:NT], self._bZ[:NT], self._bZs[:NT], self._bcost[:NT]
        for t in (self.X, self.X2, self.W, self.Ht, self.Z, self.Zs, self.cost_frames):
            t.zero_()
        self.g.fill_(1.0)
        self.B1 = None
        # (device copies of the frame tables, cached by frame structure: an upload from pageable memory would make the host wait
        # for the GPU every batch)
        key = tuple(fc)
        cache = self.__dict__.setdefault("_tab_cache", {})
        if key not in cache:
            if len(cache) >= 8:
                cache.pop(next(iter(cache)))
            cache[key] = (torch.from_numpy(self.frame_off.copy()).to(dev),
                          torch.repeat_interleave(torch.arange(self.U, dtype=torch.int32), torch.tensor(fc)).to(dev))
        self.d_frame_off, self.d_frame_utt = cache[key]
        return self

    def utt_slice(self, u):
        return slice(int(self.frame_off[u]), int(self.frame_off[u + 1]))

    def set_spectrogram(self, X):
        """X: list of complex64 (N_u, F) numpy arrays, or a device tensor [NT,Fs,2]."""
        if isinstance(X, torch.Tensor):
            self.X.copy_(X)
        else:
            Xc = np.zeros((self.NT, self.Fs), np.complex64)
            for u, x in enumerate(X):
                Xc[self.utt_slice(u), :self.F] = x
            self.X.copy_(torch.from_numpy(Xc.view(np.float32).reshape(self.NT, self.Fs, 2)))
        check(lib().vaenmf_power_spec(_ptr(self.X), _ptr(self.X2), self.NT * self.Fs, _stream()))   # mcem.py:47

    def init_nmf(self, W0, H0):
        """W0[u] (F,K), H0[u] (K,N_u) float32 (mcem.py:42-44); g = 1."""
        W = np.zeros((self.U, self.Fs, self.Kp), np.float32)
        Ht = np.zeros((self.NT, self.Kp), np.float32)
        for u in range(self.U):
            W[u, :self.F, :self.K] = W0[u]
            Ht[self.utt_slice(u), :self.K] = np.asarray(H0[u]).T
        self.W.copy_(torch.from_numpy(W))
        self.Ht.copy_(torch.from_numpy(Ht))
        self.g.fill_(1.0)

    def init_nmf_device(self, salt=0, eps=1e-8):
        """W = max(U(0,1), eps), H = max(U(0,1), eps), g = 1 (mcem.py:42-44, :51) by the library's counter-based generator,
        keyed by the utterance seeds of bind(): no torch kernel on the throughput path, and an utterance's start does not
        depend on its batch."""
        check(lib().vaenmf_init_nmf(self._plan, _ptr(self.W), _ptr(self.Ht), _ptr(self.g), C.c_uint64(int(salt) & (2 ** 64 - 1)), float(eps), _stream()))

    def dense(self, x, w, b, act):
        """act(x w^T + b) on the device (models.py:101-104 / 57-62)."""
        M, inn = x.shape
        out = w.shape[0]
        y = torch.empty(M, out, device=self.device, dtype=torch.float32)
        check(lib().vaenmf_dense(_ptr(x, rows_strided=True), M, inn, x.stride(0), _ptr(w), _ptr(b), out, act, _ptr(y), out, _stream()))
        return y

    def _dev(self, a):
        """Device copy of a host weight array, uploaded once per array object (an upload from pageable memory per call
        makes the host wait for the GPU).  CONTRACT: weight arrays handed to encode() / classify() are immutable -- the
        copy is keyed by the array object, so an array modified in place would keep its stale device copy; pass a new
        array (or call invalidate_weight_caches()) to change weights.  The drop-in classes (vaenmf.mcem) rebuild their
        engine when a model tensor's in-place version counter moves."""
        cache = self.__dict__.setdefault("_w_cache", {})
        ent = cache.get(id(a))
        if ent is None or ent[0] is not a:
            if len(cache) >= 64:
                cache.clear()
            ent = cache[id(a)] = (a, torch.from_numpy(_np32(a)).to(self.device))
        return ent[1]

    def invalidate_weight_caches(self):
        """Forget the device copies of encoder / classifier weights (after modifying weight arrays in place)."""
        self.__dict__.pop("_w_cache", None)
        self.__dict__.pop("_clf_cache", None)

    def encode(self, enc, y=None):
        """Z = posterior mean of encoder(|X|^2 [cat y]) (mcem.py:367-368 / :214-215).
        enc = [(W,b)...hidden, (Wmu,bmu)] float32 numpy."""
        x = self.X2[:, :self.F]
        if y is not None:
            x = torch.cat([x, y], dim=1).contiguous()
        t = self._dev
        h = x
        for (w, b) in enc[:-1]:
            h = self.dense(h, t(w), t(b), _lib.ACT_TANH)
        w, b = enc[-1]
        mu = self.dense(h, t(w), t(b), _lib.ACT_NONE)        # (NT, L)
        if self.L == LAT:
            self.Z.copy_(mu)
        else:                                                 # latent dimension 16: columns 16..31 are the zero padding
            self.Z.zero_()
            self.Z[:, :self.L].copy_(mu)

    def classify(self, clf, mean=None, std=None, eps=1e-8):
        """Labels from the classifier (scripts/evaluate_M2_vad.py:122-131): optional mean/std
        normalisation of |X|^2 (folded into the first layer on the host), ReLU hidden layers,
        sigmoid output, hard threshold 0.5.  clf = [(W,b) hidden..., (W,b) output] float32 numpy;
        mean/std numpy (F,1).  Returns (y_soft, y_hard) device float32 [NT,Dy]."""
        cache = self.__dict__.setdefault("_clf_cache", {})
        ent = cache.get(id(clf))
        if ent is None or ent[0] is not clf or ent[1] is not mean or ent[2] is not std:       # folded layers on the device, once per classifier
            t = lambda a: torch.from_numpy(_np32(a)).to(self.device)
            layers = [(np.asarray(w, np.float64), np.asarray(b, np.float64)) for w, b in clf]
            if mean is not None:
                m = np.asarray(mean, np.float64).reshape(-1)
                s = np.asarray(std, np.float64).reshape(-1) + eps
                w0, b0 = layers[0]
                layers[0] = (w0 / s[None, :], b0 - (w0 / s[None, :]) @ m)
            if len(cache) >= 8:
                cache.clear()
            ent = cache[id(clf)] = (clf, mean, std, [(t(w), t(b)) for w, b in layers])
        dl = ent[3]
        h = self.X2[:, :self.F]
        for (w, b) in dl[:-1]:
            h = self.dense(h, w, b, _lib.ACT_RELU)
        w, b = dl[-1]
        return self.dense(h, w, b, _lib.ACT_SIGMOID), self.dense(h, w, b, _lib.ACT_STEP)

    def set_noise_psd(self, Vb):
        """Fixed noise variance (the *_noNMF variants, mcem.py:493-760): Vb device float32 [NT,Fs] or None."""
        self._Vb_ext = None if Vb is None else Vb.to(self.device, torch.float32).contiguous()
        check(lib().vaenmf_set_noise_psd(self._plan, _ptr(self._Vb_ext)))

    def set_labels(self, y):
        """y device float32 [NT,Dy]: fold the label half of the first decoder layer."""
        if self.Dy == 0:
            raise ValueError("decoder has no label input")
        y = y.to(self.device, torch.float32).contiguous()
        if getattr(self, "_bB1", None) is None:           # fixed address from batch to batch (vaenmf_em_run's graph signature)
            self._bB1 = torch.empty(self._max_frames, HID, device=self.device, dtype=torch.float32)
        self.B1 = self._bB1[:self.NT]
        check(lib().vaenmf_layer1_bias(self._plan, _ptr(y), self.Dy, _ptr(self.B1), _stream()))

    # ------------------------------------------------------------------ hot path
    def mh_chain(self, nsamples, burnin, var_rw, call=0, eps=None, u=None, want_acc=False, update_Z=True):
        rng = _lib.Rng(_lib.RNG_DEVICE if eps is None else _lib.RNG_REPLAY, int(call), _ptr(eps), _ptr(u))
        acc = torch.empty(nsamples + burnin, self.NT, device=self.device, dtype=torch.float32) if want_acc else None
        check(lib().vaenmf_mh_chain(self._plan, _ptr(self.X2), _ptr(self.W), _ptr(self.Ht), _ptr(self.g), _ptr(self.Z),
                                    int(bool(update_Z)), _ptr(self.B1), _ptr(self.Zs), self.Rcap, int(nsamples), int(burnin), float(var_rw),
                                    C.byref(rng), _ptr(acc), _stream()))
        return acc

    def sample_store(self, on=True):
        """Keep the decoded variances of the chain's samples in HBM (see include/vaenmf.h); sized here, for the
        bound batch and chains of up to Rcap samples per frame (vaenmf_mh_chain itself never allocates)."""
        check(lib().vaenmf_sample_store(self._plan, int(self.Rcap) if on else 0))

    def stored_variances(self, R):
        """Vs of the last chain's samples from the store: device float32 [NT,R,Fs]."""
        out = torch.empty(self.NT, R, self.Fs, device=self.device, dtype=torch.float32)
        check(lib().vaenmf_sample_store_gather(self._plan, _ptr(out), _stream()))
        return out

    def rng_fill(self, call, S):
        eps = torch.empty(S, self.NT, LAT, device=self.device, dtype=torch.float32)
        u = torch.empty(S, self.NT, device=self.device, dtype=torch.float32)
        check(lib().vaenmf_rng_fill(self._plan, int(call), int(S), _ptr(eps), _ptr(u), _stream()))
        return eps, u

    def decode(self, R):
        Vs = torch.empty(self.NT, R, self.Fs, device=self.device, dtype=torch.float32)
        check(lib().vaenmf_decode(self._plan, _ptr(self.Zs), self.Rcap, int(R), _ptr(self.B1), _ptr(Vs), _stream()))
        return Vs

    def m_step(self, R):
        check(lib().vaenmf_m_step(self._plan, _ptr(self.X2), _ptr(self.W), _ptr(self.Ht), _ptr(self.g), _ptr(self.Zs),
                                  self.Rcap, int(R), _ptr(self.B1), _ptr(self.cost_frames), _stream()))
        return self.cost_frames

    def m_step_stored(self):
        """M-step over the sample store of the last chain (see include/vaenmf.h)."""
        check(lib().vaenmf_m_step_stored(self._plan, _ptr(self.X2), _ptr(self.W), _ptr(self.Ht), _ptr(self.g),
                                         _ptr(self.cost_frames), _stream()))
        return self.cost_frames

    def wiener_stored(self, want_masks=False):
        S = torch.empty_like(self.X)
        N = torch.empty_like(self.X)
        WFs = torch.empty(self.NT, self.Fs, device=self.device, dtype=torch.float32) if want_masks else None
        WFn = torch.empty(self.NT, self.Fs, device=self.device, dtype=torch.float32) if want_masks else None
        check(lib().vaenmf_wiener_stored(self._plan, _ptr(self.W), _ptr(self.Ht), _ptr(self.g), _ptr(self.X), _ptr(S), _ptr(N),
                                         _ptr(WFs), _ptr(WFn), _stream()))
        return S, N, WFs, WFn

    def wiener(self, R, want_masks=False):
        S = torch.empty_like(self.X)
        N = torch.empty_like(self.X)
        WFs = torch.empty_like(self.X2) if want_masks else None
        WFn = torch.empty_like(self.X2) if want_masks else None
        check(lib().vaenmf_wiener(self._plan, _ptr(self.X2), _ptr(self.W), _ptr(self.Ht), _ptr(self.g), _ptr(self.Zs),
                                  self.Rcap, int(R), _ptr(self.B1), _ptr(self.X), _ptr(S), _ptr(N), _ptr(WFs), _ptr(WFn), _stream()))
        return S, N, WFs, WFn

    def run(self, niter, nsE, biE, nsWF, biWF, var_rw, store=None):
        """Fused EM.run for the whole batch (device RNG).  Returns (cost [U,niter] float64, S_hat, N_hat).
        store: use the sample-variance store (include/vaenmf.h); default: on (bf16 rows in bf16 mode, float rows in
        bf16x3 mode; vaenmf_em_run falls back to the decoding M-step when a batch's store would pass 3.5 GB --
        VAENMF_Q_MSTEP_PATH tells which path ran)."""
        if int(niter) not in self._bcu:
            self._bcu[int(niter)] = torch.empty(self._max_utts * int(niter), device=self.device, dtype=torch.float64)
        cost = self._bcu[int(niter)][:self.U * int(niter)].view(self.U, int(niter))
        cost.zero_()
        S, N = self._bS[:self.NT], self._bN[:self.NT]
        if store is None:       # the chain keeps the samples' variances in HBM, M-step and Wiener filter stream them
            store = True
        self.sample_store(store)
        check(lib().vaenmf_em_run(self._plan, _ptr(self.X2), _ptr(self.W), _ptr(self.Ht), _ptr(self.g), _ptr(self.Z),
                                  _ptr(self.B1), _ptr(self.Zs), self.Rcap, int(niter), int(nsE), int(biE), int(nsWF),
                                  int(biWF), float(var_rw), _ptr(self.X), _ptr(S), _ptr(N), _ptr(cost), _stream()))
        self.sample_store(False)
        return cost.clone(), S.clone(), N.clone()      # (the buffers are overwritten by the next run)

    # ------------------------------------------------------------------ host views (reference shapes)
    def Vb(self, u):
        sl = self.utt_slice(u)
        return (self.W[u, :self.F, :self.K] @ self.Ht[sl, :self.K].T)          # (F,N) mcem.py:82

    def cost_from_frames(self, R):
        """mean over (R,F,N) per utterance of the per-frame sums (mcem.py:70)."""
        cf = self.cost_frames.cpu().numpy()
        return np.array([cf[self.utt_slice(u)].sum() / (R * self.F * (self.frame_off[u + 1] - self.frame_off[u]))
                         for u in range(self.U)])

"""torch.autograd.Function wrappers around the HIP kernels (hip_ops).

Each Function mirrors one ATen-level operator group of the reference hot path (SURVEY.md
section 2.2) with an explicit backward; autograd is only the glue that chains them.
"""
import numpy as np
import torch

from . import hip_ops as H


def _pad4(n):
    return H.round_up(n, 4)


# ---- hand-offs between neighbouring Functions of the tail --------------------------------------------------
# head (final Linear) -> fused mask / iSTFT -> LogMAE are three autograd Functions (three reference modules:
# net.py:629-666, enhancer.py:98-100 + model.py:661-664, loss.py:244-247), but on the training step the
# backward of the middle one can take the loss's arguments instead of a [B,K,N] gradient and write d(logit)
# where the Linear's backward reads it.  A producer that folds its work into its neighbour returns _dummy_grad
# (zeros without memory) and leaves the real arguments on a link object both Functions hold; a consumer that
# receives anything but that pristine dummy (another loss used the same tensor, autograd summed gradients)
# falls back to the unfused kernels for the linked part and adds it -- correct either way.
_ZERO = {}


def _dummy_grad(like, shape=None):
    z = _ZERO.get(like.device)
    if z is None:
        z = _ZERO[like.device] = torch.zeros(1, device=like.device, dtype=torch.float32)
    return z.expand(tuple(like.shape if shape is None else shape))


def _is_dummy(g):
    z = _ZERO.get(g.device)
    return z is not None and g.data_ptr() == z.data_ptr() and all(st == 0 for st in g.stride())


class _Link:
    """payload: set by the producer's backward, taken (and cleared) by the consumer's."""

    def __init__(self, **kw):
        self.payload = None
        self.__dict__.update(kw)

    def take(self):
        p, self.payload = self.payload, None
        return p

    def observed(self):
        """True when somebody watches the gradient of the tensor this link hangs on (``register_hook``,
        ``retain_grad``): a fold that never forms that gradient must not be taken then."""
        ref = getattr(self, "out", None)
        t = ref() if ref is not None else None
        return t is not None and (bool(getattr(t, "_backward_hooks", None)) or bool(getattr(t, "retains_grad", False)))


# ------------------------------------------------------------------------------ RNNP
class _RNNP(torch.autograd.Function):
    """One RNNP_packed layer (tssep/train/rnnp.py:88-96,146-168): BLSTM + Linear (+ tanh).

    x: [R, ld_x] buffer view, rows (n, t) for N sequences of T frames, I valid columns.
    Output: [R, hdim] rows (padded leading dimension), or the speaker-combined layout
    [B, T, K*hdim] when ``combine=K`` (net.py:608-611 fused into the projection's store)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r, w_proj, b_proj,
                N, T, act, combine, in_tanh=0, in_link=None, out_link=None):
        dev = x.device
        Hh = w_hh.shape[1]
        I = w_ih.shape[1]
        hdim = w_proj.shape[0]
        xv, ld_x = H.rows_view(x)
        R = N * T
        assert xv.shape[0] == R and xv.shape[1] >= I, (xv.shape, R, I)
        lstm_params = [w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r]
        pk = dict(H.derived("lstm_pack", lstm_params, lambda: H.lstm_pack(lstm_params, Hh, I)))
        gates = torch.empty(R, 8 * Hh, device=dev, dtype=torch.float32)
        H.gemm(xv, ld_x, pk["wih_p"], pk["ld_i"], gates, 8 * Hh, R, 8 * Hh, I, bias=pk["bias_p"])
        Hp = _pad4(Hh)
        cell = torch.empty(N, T, 2, Hh, device=dev, dtype=torch.float32)
        hout = (torch.zeros if Hp != Hh else torch.empty)(R, 2 * Hp, device=dev, dtype=torch.float32)
        kf, kb = H.recurrence_kernel(N, Hh, False, T, dev), H.recurrence_kernel(N, Hh, True, T, dev)
        cf = cb = wf3 = wb3 = None
        if "cluster" in (kf, kb):
            cf, cb = H.derived("pack_cluster", [w_hh, w_hh_r], lambda: H.lstm_pack_cluster(w_hh, w_hh_r, Hh))
        # the 32-sequence W-stationary kernels' packed weights only where one of the two directions of time runs on them
        g16 = H.onchip16_groups(N, Hh, dev) if kf == "onchip" and (2 * Hp) % 4 == 0 and Hp % 4 == 0 else 0
        g16b = H.onchip16_bwd_groups(N, Hh, dev) if kb == "onchip" and Hp % 4 == 0 else 0
        if (kf == "onchip" and not g16) or (kb == "onchip" and not g16b):
            wf3, wb3 = H.derived("pack_onchip", [w_hh, w_hh_r], lambda: H.lstm_pack_onchip(w_hh, w_hh_r, Hh))
        if kf == "cluster":
            H.blstm_cluster_fwd(gates, cell, hout, 2 * Hp, Hp, cf, N, T, Hh)
        elif kf == "onchip":
            if g16:      # interleaved 16-sequence groups (round 3)
                wf16 = H.derived("pack_onchip16", [w_hh, w_hh_r], lambda: H.lstm_pack_onchip16(w_hh, w_hh_r, Hh))
                H.blstm_onchip16_fwd(gates, cell, hout, 2 * Hp, Hp, wf16, N, T, Hh, g16)
            else:
                H.blstm_onchip_fwd(gates, cell, hout, 2 * Hp, Hp, wf3, N, T, Hh)
        else:
            H.blstm_fwd(gates, cell, hout, 2 * Hp, Hp, pk["whh_f"], N, T, Hh)
        pk["whh_cb"] = cb if kb == "cluster" else None
        pk["whh_ob"] = wb3 if kb == "onchip" else None
        pk["bwd_onchip"] = kb == "onchip"
        # projection weight in the (possibly padded) [hdim, 2*Hp] column layout of hout
        wp = H.derived("proj_layout", [w_proj], lambda: _proj_layout(w_proj, Hh, Hp))
        if combine:
            K = combine
            B = N // K
            y = torch.empty(B * T, K * hdim, device=dev, dtype=torch.float32)
            H.gemm(hout, 2 * Hp, wp, 2 * Hp, y, 0, R, hdim, 2 * Hp, bias=b_proj.detach(), act=act,
                   remap=dict(T=T, K=K, sb=T * K * hdim, sk=hdim, st=K * hdim))
            ld_y = K * hdim
        else:
            y, ld_y = H.padded(R, hdim, dev, zero=True)
            H.gemm(hout, 2 * Hp, wp, 2 * Hp, y, ld_y, R, hdim, 2 * Hp, bias=b_proj.detach(), act=act)
        ctx.save_for_backward(xv, gates, cell, hout, y, wp)
        ctx.params = (w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r, w_proj, b_proj)
        ctx.pk = pk
        ctx.meta = (N, T, I, Hh, Hp, hdim, ld_x, ld_y, act, combine)
        # Tanh-backward fold (see rnnp_layer): in_link = the link my producer hung on x, out_link = the one my
        # consumer may answer on.  Both are decided again in backward, when it is known who else uses the tensors.
        ctx.fold = (int(in_tanh) if in_link is not None else 0, in_link, out_link)
        assert out_link is None or act == 1, "out_link: only behind the fused Tanh"
        ctx.x_shape = x.shape
        return y if combine else y[:, :hdim]

    @staticmethod
    def backward(ctx, dy):
        xv, gates, cell, hout, y, wp = ctx.saved_tensors
        N, T, I, Hh, Hp, hdim, ld_x, ld_y, act, combine = ctx.meta
        pk = ctx.pk
        dev = dy.device
        R = N * T
        K = combine if combine else 1
        # d(pre-activation) of the projection, rows (n,t) x hdim, contiguous
        in_tanh, in_link, out_link = ctx.fold
        folded = out_link.take() if out_link is not None else None
        if folded is not None:
            # the consumer's d(input) GEMM already applied 1 - y^2 and wrote dense rows (n,t) x hdim; `dy` is the
            # dummy it returned -- unless y has ANOTHER consumer (auxiliary loss): autograd then summed that
            # consumer's plain d(y) onto the dummy's zeros, and it goes through the Tanh backward here
            dz = folded.view(R, hdim)
            if not _is_dummy(dy):
                dyc = dy.contiguous() if combine else _dense_rows(dy, hdim)
                yc = y if combine else _dense_rows(y[:, :hdim], hdim)
                dz = dz + H.tanh_bwd(dyc, yc, R, hdim, K, T, bool(combine))
        elif act:
            dyc = dy.contiguous() if combine else _dense_rows(dy, hdim)
            yc = y if combine else _dense_rows(y[:, :hdim], hdim)
            dz = H.tanh_bwd(dyc, yc, R, hdim, K, T, bool(combine))
        else:
            if combine:   # layout change only: reuse tanh_bwd's gather with y = 0
                dz = H.tanh_bwd(dy.contiguous(), torch.zeros_like(dy), R, hdim, K, T, True)
            else:
                # (a strided [R, hdim] view of a padded buffer -- what the conditioning's backward hands the pre-net --
                # goes to the GEMMs as it is: rows_view only copies what is not 16-byte-row addressable)
                dz = dy if dy.dim() == 2 else dy.reshape(-1, hdim)
        dz, ld_dz = H.rows_view(dz)
        G = 8 * Hh
        params = ctx.params
        sinks = [_grad_sink(p) for p in params]
        direct = all(s_ is not None for s_ in sinks) and H.OVERLAP_WGRAD
        main = torch.cuda.current_stream()
        side = H.side_stream(dev, R) if direct else main

        # ---- projection weight / bias gradients (side stream when direct)
        def proj_wgrads():
            if direct and Hp == Hh and H.fused_colsum():
                # the bias gradient (column sums of dz) rides on the weight-gradient GEMM as a virtual ones
                # column of hout: one pass over dz instead of two, and N = 600 -> 601 costs no extra tile
                part, S = H.wgrad(dz, ld_dz, hout, 2 * Hp, hdim, 2 * Hp, R, with_colsum=True)
                H.reduce_splits_bias(part, S, hdim, 2 * Hp, H.round_up(2 * Hp + 1, 4), sinks[8], sinks[9],
                                     accumulate=True)
                return None, None
            part, S = H.wgrad(dz, ld_dz, hout, 2 * Hp, hdim, 2 * Hp, R)
            if direct and Hp == Hh:
                H.reduce_splits(part, S, hdim * 2 * Hp, sinks[8], accumulate=True)
                H.colsum(dz, ld_dz, R, hdim, out=sinks[9], accumulate=True)
                return None, None
            dwp = torch.empty(hdim, 2 * Hp, device=dev, dtype=torch.float32)
            H.reduce_splits(part, S, hdim * 2 * Hp, dwp)
            d_w = _proj_unlayout(dwp, Hh, Hp)
            d_b = H.colsum(dz, ld_dz, R, hdim)
            if direct:
                sinks[8].add_(d_w)
                sinks[9].add_(d_b)
                return None, None
            return d_w, d_b

        if direct:
            side.wait_stream(main)
            for t_ in (dz, hout):
                t_.record_stream(side)
            with torch.cuda.stream(side):
                d_w_proj, d_b_proj = proj_wgrads()
        else:
            d_w_proj, d_b_proj = proj_wgrads()
        # ---- critical path: dhout, BPTT (gates <- d pre-activations)
        dhout = torch.empty(R, 2 * Hp, device=dev, dtype=torch.float32)
        w_proj = params[8]      # (the builds read the parameters, not this step's tensors: hip_ops.prepare_derived)
        wpT, ld_t = H.derived("proj_T", [w_proj], lambda: H.transposed(
            H.derived("proj_layout", [w_proj], lambda: _proj_layout(w_proj, Hh, Hp)), hdim, 2 * Hp))
        H.gemm(dz, ld_dz, wpT, ld_t, dhout, 2 * Hp, R, 2 * Hp, hdim)
        if pk.get("whh_cb") is not None:
            H.blstm_cluster_bwd(gates, cell, dhout, 2 * Hp, Hp, pk["whh_cb"], N, T, Hh)
        elif pk.get("bwd_onchip"):
            g16 = H.onchip16_bwd_groups(N, Hh, gates.device) if Hp % 4 == 0 else 0
            if g16:      # interleaved 16-sequence groups (round 3)
                w_hh, w_hh_r = ctx.params[1], ctx.params[5]
                wb16 = H.derived("pack_onchip16_bwd", [w_hh, w_hh_r], lambda: H.lstm_pack_onchip16_bwd(w_hh, w_hh_r, Hh))
                H.blstm_onchip16_bwd(gates, cell, dhout, 2 * Hp, Hp, wb16, N, T, Hh, g16)
            else:
                H.blstm_onchip_bwd(gates, cell, dhout, 2 * Hp, Hp, pk["whh_ob"], N, T, Hh)
        else:
            H.blstm_bwd(gates, cell, dhout, 2 * Hp, Hp, pk["whh_b"], N, T, Hh)

        # ---- LSTM weight / bias gradients from dgates (side stream when direct)
        def lstm_wgrads():
            dwhh = torch.empty(2, 4 * Hh * Hh, device=dev, dtype=torch.float32)
            for d in range(2):   # dgates_t paired with h_{t-1} (forward dir) / h_{t+1} (reverse)
                part, S = H.wgrad((gates, d * 4 * Hh), G, (hout, d * Hp), 2 * Hp, 4 * Hh, Hh, R,
                                  b_kshift=(-1 if d == 0 else 1), kperiod=T)
                H.reduce_splits(part, S, 4 * Hh * Hh, dwhh[d])
            # dW_ih and (split-bf16 GEMM) the bias gradient in ONE pass over dgates: a virtual ones
            # column of x makes column I of the partials the column sums
            fused = H.fused_colsum()
            part, S = H.wgrad(gates, G, xv, ld_x, G, I, R, with_colsum=fused)
            ldp = H.round_up(I + 1, 4) if fused else I
            if fused:
                cs, cs_ld, cs_S, cs_stride = (part, I), ldp, S, G * ldp
            else:
                cs, cs_ld, cs_S, cs_stride = H.colsum(gates, G, R, G), 1, 1, 0
            if direct:
                H.lstm_unpack(dwhh, Hh, 1, 0, Hh, Hh, sinks[1], sinks[5], accumulate=True)
                H.lstm_unpack(part, ldp, S, G * ldp, Hh, I, sinks[0], sinks[4], accumulate=True)
                for a_, b_ in ((2, 6), (3, 7)):      # b_ih and b_hh receive the same gradient
                    H.lstm_unpack(cs, cs_ld, cs_S, cs_stride, Hh, 1, sinks[a_], sinks[b_], accumulate=True)
                return (None,) * 8
            new = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)  # noqa: E731
            d_whh, d_whh_r = new(4 * Hh, Hh), new(4 * Hh, Hh)
            H.lstm_unpack(dwhh, Hh, 1, 0, Hh, Hh, d_whh, d_whh_r)
            d_wih, d_wih_r = new(4 * Hh, I), new(4 * Hh, I)
            H.lstm_unpack(part, ldp, S, G * ldp, Hh, I, d_wih, d_wih_r)
            d_b, d_b_r = new(4 * Hh), new(4 * Hh)
            H.lstm_unpack(cs, cs_ld, cs_S, cs_stride, Hh, 1, d_b, d_b_r)
            return d_wih, d_whh, d_b, d_b.clone(), d_wih_r, d_whh_r, d_b_r, d_b_r.clone()

        if direct:
            side.wait_stream(main)
            for t_ in (gates, hout, xv):
                t_.record_stream(side)
            with torch.cuda.stream(side):
                lstm_grads = lstm_wgrads()
            _notify_grads(params)                    # (10 tensors: both directions' LSTM weights + the projection)
        else:
            lstm_grads = lstm_wgrads()
        dx = None
        if ctx.needs_input_grad[0]:
            lstm_params = list(params[:8])

            def wih_t():
                pk_ = H.derived("lstm_pack", lstm_params, lambda: H.lstm_pack(lstm_params, Hh, I))
                return H.transposed(pk_["wih_p"].view(G, pk_["ld_i"]), G, I)
            wihT, ld_t = H.derived("wih_T", [params[0], params[4]], wih_t)
            if in_tanh and not in_link.observed():
                # my input is a Tanh output: its backward rides on this GEMM's store and the result travels on
                # the link; x receives a dummy (its true gradient is never formed -- hence not when it is watched)
                Kc = in_tanh
                assert I % 4 == 0 and I % Kc == 0 and ld_x == I, (I, Kc, ld_x)
                dxb = torch.empty(R, I, device=dev, dtype=torch.float32)
                remap = None
                if Kc > 1:    # rows (b,t) x (k, hdim) -> rows (b,k,t) x hdim (inverse of net.py:608-611)
                    hd = I // Kc
                    remap = dict(T=T, K=1, sb=Kc * T * hd, sk=0, st=hd, cm=hd, co=T * hd)
                H.gemm(gates, G, wihT, ld_t, dxb, 0 if remap else I, R, I, G, act=2, aux=(xv, ld_x), remap=remap)
                # (accumulated, like head_link below: the activation may feed TWO folded consumers -- ADVICE r4)
                in_link.payload = dxb if in_link.payload is None else in_link.payload + dxb.view_as(in_link.payload)
                dx = _dummy_grad(xv, ctx.x_shape)
            else:
                dxb, ld_dx = H.padded(R, I, dev, zero=True)
                H.gemm(gates, G, wihT, ld_t, dxb, ld_dx, R, I, G)
                dx = dxb[:, :I]
                if tuple(ctx.x_shape) != tuple(dx.shape):
                    dx = dx.reshape(ctx.x_shape)
        return (dx, *lstm_grads, d_w_proj, d_b_proj, None, None, None, None, None, None, None)


def _notify_grads(params):
    """The gradients of `params` are queued completely (directly into the flat bucket): a bucket with per-layer
    segments may all-reduce the layer now (distributed.GradBucket.notify; a no-op unless armed)."""
    b = getattr(params[0], "_tssep_bucket", None)
    b = b() if b is not None else None
    if b is not None and b._armed:
        b.notify(params)


def _grad_sink(p):
    """Flat-bucket view a parameter's gradient may be accumulated into directly (set by
    tssep_amd.distributed.GradBucket), or None -> return the gradient through autograd."""
    sinks = getattr(p, "_tssep_grad_sinks", None)
    if not sinks or p.grad is None or p.grad.data_ptr() != sinks[0].data_ptr():
        return None
    return sinks[H.ACTIVE_SINK % len(sinks)]


def _dense_rows(t, cols):
    t = t if t.dim() == 2 else t.reshape(-1, cols)
    return t.contiguous()


def _proj_layout(w_proj, Hh, Hp):
    w = w_proj.detach()
    if Hp == Hh and w.is_contiguous() and w.data_ptr() % 16 == 0:
        return w
    out = torch.zeros(w.shape[0], 2 * Hp, device=w.device, dtype=torch.float32)
    out[:, :Hh] = w[:, :Hh]
    out[:, Hp:Hp + Hh] = w[:, Hh:]
    return out


def _proj_unlayout(dwp, Hh, Hp):
    if Hp == Hh:
        return dwp
    return torch.cat([dwp[:, :Hh], dwp[:, Hp:Hp + Hh]], dim=1).contiguous()


def rnnp_layer(x, lstm, linear, N, T, act=0, combine=0, in_tanh=0, dz_given=False):
    """x rows (n,t); lstm = torch.nn.LSTM parameter container, linear = nn.Linear container.
    The Tanh between two layers (net.py:623-625) runs forward in the producer's projection epilogue (act = 1);
    its BACKWARD may run in the store of the consumer's d(input) GEMM: the producer is called with dz_given and
    hangs a _Link on its output, the consumer with in_tanh (1: same row layout; K > 1: its input is the
    producer's speaker-combined tensor [B T, K hdim] and the gradient is stored back as rows (b,k,t) x hdim).
    In backward the consumer leaves d(pre-activation) on the link and returns a dummy for its input; the
    producer adds the Tanh backward of whatever ELSE arrived for its output (a second consumer), and when the
    output is watched (hook / retain_grad) or the consumer never ran, nothing is folded and `tssep_tanh_bwd`
    runs as without the fold -- correct either way."""
    import weakref
    in_link = getattr(x, "_tssep_tanh_link", None) if in_tanh else None
    if in_link is not None and not (x.requires_grad and torch.is_grad_enabled()):
        in_link = None
    out_link = _Link() if (dz_given and act == 1 and torch.is_grad_enabled()) else None
    out = _RNNP.apply(x, lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0,
                      lstm.weight_ih_l0_reverse, lstm.weight_hh_l0_reverse,
                      lstm.bias_ih_l0_reverse, lstm.bias_hh_l0_reverse,
                      linear.weight, linear.bias, N, T, act, combine, in_tanh, in_link, out_link)
    if out_link is not None and out.requires_grad:
        out_link.out = weakref.ref(out)
        out._tssep_tanh_link = out_link
    return out


# ---------------------------------------------------------------------- conditioning
class _Cond(torch.autograd.Function):
    """tssep/train/net.py:862-896 (+ trial fold :913-924).  pre rows (b,t) -> rows (b,tr,k,t)."""

    @staticmethod
    def forward(ctx, pre, aux, B, K, T, trials, combination):
        F = pre.shape[-1]
        pv, ld_pre = H.rows_view(pre)
        xs, ld, info = H.cond_fwd(pv, ld_pre, aux.detach(), B, K, T, F, trials, combination)
        W = F if combination == "mul" else F + aux.shape[-1]
        ctx.info = info
        ctx.meta = (B, K, T, F, trials, combination, ld)
        return xs[:, :W]

    @staticmethod
    def backward(ctx, dxs):
        B, K, T, F, trials, combination, ld = ctx.meta
        dv, ld_d = H.rows_view(dxs)
        dpre, ldp = H.cond_bwd(dv, ld_d, ctx.info, B, K, T, F, trials, combination)
        return dpre[:, :F], None, None, None, None, None, None


def condition(pre, aux, B, K, T, trials, combination):
    return _Cond.apply(pre, aux, B, K, T, trials, combination)


# ------------------------------------------------------------------------ final linear
class _Head(torch.autograd.Function):
    """post_net.linear + final einops + trial mean + speaker un-permutation
    (tssep/train/net.py:629-666, 928-967): x rows -> logit [B,K,T,F]."""

    @staticmethod
    def forward(ctx, x, weight, bias, perm, iperm, B, K, T, F, trials, Fr, spk_rows, link=None):
        dev = x.device
        ctx.link = link
        xv, ld_x = H.rows_view(x)
        P = weight.shape[1]
        wv, ld_w = H.rows_view(weight.detach())
        R = xv.shape[0]
        Nout = weight.shape[0]
        fast = (not spk_rows) and trials == 1 and Fr == F
        if fast:      # store straight into [B, perm[k], T, F] from the GEMM epilogue
            out = torch.empty(B, K, T, F, device=dev, dtype=torch.float32)
            H.gemm(xv, ld_x, wv, ld_w, out, 0, R, Nout, P, bias=bias.detach(),
                   remap=dict(T=T, K=1, sb=K * T * F, sk=0, st=F, cm=F, co=T * F, perm=perm,
                              perm_ld=K))
        else:
            raw = torch.empty(R, Nout, device=dev, dtype=torch.float32)
            H.gemm(xv, ld_x, wv, ld_w, raw, Nout, R, Nout, P, bias=bias.detach())
            out = H.logit_map_fwd(raw, perm, iperm, B, trials, K, T, F, Fr, spk_rows)
        ctx.save_for_backward(xv, wv)
        ctx.params = (weight, bias)
        ctx.aux = (perm, iperm)
        ctx.meta = (B, K, T, F, trials, Fr, spk_rows, ld_x, ld_w, P, R, Nout)
        ctx.x_shape = x.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        xv, wv = ctx.saved_tensors
        perm, iperm = ctx.aux
        B, K, T, F, trials, Fr, spk_rows, ld_x, ld_w, P, R, Nout = ctx.meta
        dev = dout.device
        draw = ctx.link.take() if ctx.link is not None else None
        if draw is None:
            draw = H.logit_map_bwd(dout, perm, iperm, B, trials, K, T, F, Fr, spk_rows).view(R, Nout)
        elif not _is_dummy(dout):      # the fused tail wrote its part already laid out; someone else used logit too
            draw = draw + H.logit_map_bwd(dout, perm, iperm, B, trials, K, T, F, Fr, spk_rows).view(R, Nout)
        dv, ld_d = H.rows_view(draw)
        sw, sb = _grad_sink(ctx.params[0]), _grad_sink(ctx.params[1])
        direct = sw is not None and sb is not None and H.OVERLAP_WGRAD
        if direct:
            main, side = torch.cuda.current_stream(), H.side_stream(dev, R)
            side.wait_stream(main)
            for t_ in (dv, xv):
                t_.record_stream(side)
            with torch.cuda.stream(side):
                if H.fused_colsum():
                    part, S = H.wgrad(dv, ld_d, xv, ld_x, Nout, P, R, with_colsum=True)
                    H.reduce_splits_bias(part, S, Nout, P, H.round_up(P + 1, 4), sw, sb, accumulate=True)
                else:
                    part, S = H.wgrad(dv, ld_d, xv, ld_x, Nout, P, R)
                    H.reduce_splits(part, S, Nout * P, sw, accumulate=True)
                    H.colsum(dv, ld_d, R, Nout, out=sb, accumulate=True)
            dw = db = None
            _notify_grads(ctx.params)
        else:
            part, S = H.wgrad(dv, ld_d, xv, ld_x, Nout, P, R)
            dw = torch.empty(Nout, P, device=dev, dtype=torch.float32)
            H.reduce_splits(part, S, Nout * P, dw)
            db = H.colsum(dv, ld_d, R, Nout)
        dxb, ld_dx = H.padded(R, P, dev, zero=True)
        weight = ctx.params[0]
        wvT, ld_t = H.derived("head_T", [weight], lambda: H.transposed(H.rows_view(weight.detach())[0], Nout, P))
        H.gemm(dv, ld_d, wvT, ld_t, dxb, ld_dx, R, P, Nout)
        dx = dxb[:, :P]
        if tuple(ctx.x_shape) != tuple(dx.shape):
            dx = dx.reshape(ctx.x_shape)
        return dx, dw, db, None, None, None, None, None, None, None, None, None, None


def head(x, linear, perm, iperm, B, K, T, F, trials, Fr, spk_rows):
    link = None
    if H.FOLD_TAIL and (not spk_rows) and trials == 1 and Fr == F and x.requires_grad:
        link = _Link(iperm=iperm, shape=(B, K, T, F))       # the layout the fused tail may write d(logit) in
    out = _Head.apply(x, linear.weight, linear.bias, perm, iperm, B, K, T, F, trials, Fr, spk_rows, link)
    if link is not None:
        out._tssep_head_link = link
    return out


# --------------------------------------------------------------------------- mask head
class _MaskHead(torch.autograd.Function):
    """sigmoid + Masking (tssep/train/net.py:981-986, tssep/train/enhancer.py:98-100)."""

    @staticmethod
    def forward(ctx, logit, obs):
        mask, est = H.maskhead_fwd(logit, obs)
        ctx.save_for_backward(mask, obs)
        return mask, est

    @staticmethod
    def backward(ctx, dmask, dest):
        mask, obs = ctx.saved_tensors
        if dest is None:
            dest = torch.zeros(mask.shape, device=mask.device, dtype=torch.complex64)
        return H.maskhead_bwd(dest, dmask, mask, obs), None


def mask_head(logit, obs):
    """logit [B,K,T,F], obs complex [B,T,F] -> (mask, stft_estimate)."""
    return _MaskHead.apply(logit, obs)


class _Sigmoid(torch.autograd.Function):
    """mask only (no observation available): reuses the mask-head kernel with obs = 0."""

    @staticmethod
    def forward(ctx, logit):
        B, K, T, F = logit.shape
        obs = torch.zeros(B, T, F, device=logit.device, dtype=torch.complex64)
        mask, _ = H.maskhead_fwd(logit, obs)
        ctx.save_for_backward(mask, obs)
        return mask

    @staticmethod
    def backward(ctx, dmask):
        mask, obs = ctx.saved_tensors
        dest = torch.zeros(mask.shape, device=mask.device, dtype=torch.complex64)
        return H.maskhead_bwd(dest, dmask, mask, obs)


def sigmoid(logit):
    return _Sigmoid.apply(logit)


# ------------------------------------------------------------------------- STFT family
_WINDOWS = {}


def windows(window, size, shift, device, window_length=None):
    """(analysis window, biorthogonal synthesis window), both [size] fp32 on `device`.  window_length < size
    (paderbox stft / istft: the window covers the first window_length samples of a frame, the transform is
    zero-padded to size): both windows are zero beyond window_length, so the kernels -- which always move `size`
    samples per frame -- compute exactly irfft(X)[..., :window_length] * w_syn."""
    from scipy.signal import get_window
    wl = size if window_length is None else int(window_length)
    key = (window, size, shift, str(device), wl)
    if key not in _WINDOWS:
        w = get_window(window, wl, fftbins=True).astype(np.float64)
        denom = np.zeros(wl)
        for i in range(-(wl // shift) - 1, wl // shift + 2):
            off = i * shift
            lo, hi = max(0, off), min(wl, wl + off)
            if lo < hi:
                denom[lo:hi] += (w ** 2)[lo - off:hi - off]
        wa, ws = np.zeros(size), np.zeros(size)
        wa[:wl] = w
        ws[:wl] = w / denom
        _WINDOWS[key] = (torch.as_tensor(wa, dtype=torch.float32).to(device),
                         torch.as_tensor(ws, dtype=torch.float32).to(device))
    return _WINDOWS[key]


class _ISTFT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, wsyn, N, size, shift, fading):
        lead = X.shape[:-2]
        T = X.shape[-2]
        y, _ = H.istft_fwd(X.reshape(-1, T, X.shape[-1]), wsyn, N, size, shift, fading)
        ctx.save_for_backward(wsyn)
        ctx.meta = (lead, T, N, size, shift, fading)
        return y.reshape(*lead, N)

    @staticmethod
    def backward(ctx, dy):
        (wsyn,) = ctx.saved_tensors
        lead, T, N, size, shift, fading = ctx.meta
        dX = H.istft_bwd(dy.reshape(-1, N), wsyn, T, size, shift, fading)
        return dX.reshape(*lead, T, size // 2 + 1), None, None, None, None, None


def istft(X, wsyn, N, size=1024, shift=256, fading=True):
    return _ISTFT.apply(X, wsyn, N, size, shift, fading)


class _MaskISTFT(torch.autograd.Function):
    """sigmoid (net.py:983) -> Masking (enhancer.py:98-100) -> fe.istft (model.py:661-664) as ONE kernel
    each way; with ``tgt`` the forward also leaves the per-chunk sums of |estimate - tgt| on the result
    (``_tssep_absdiff``), which ``log_mae`` / ``mae`` consume instead of re-reading both signals."""

    @staticmethod
    def forward(ctx, logit, obs, wsyn, N, size, shift, fading, tgt, head_link=None, loss_link=None):
        y, part = H.mask_istft_fwd(logit, obs, wsyn, N, size, shift, fading, tgt)
        ctx.save_for_backward(logit, obs, wsyn)
        ctx.meta = (size, shift, fading)
        ctx.links = (head_link, loss_link)
        ctx.mark_non_differentiable(*([part] if part is not None else []))
        return (y, part) if part is not None else (y,)

    @staticmethod
    def backward(ctx, dy, *_):
        logit, obs, wsyn = ctx.saved_tensors
        size, shift, fading = ctx.meta
        head_link, loss_link = ctx.links
        loss = loss_link.take() if loss_link is not None else None      # (est, tgt, sums, gout) of LogMAE / MAE
        if loss is not None and not _is_dummy(dy):
            dy = dy + H.logmae_bwd(loss[0].contiguous(), loss[1].contiguous(), loss[2], loss[3])
            loss = None
        if head_link is not None and tuple(head_link.shape) == tuple(logit.shape):
            d = H.mask_istft_bwd(dy, logit, obs, wsyn, size, shift, fading, loss=loss, iperm=head_link.iperm,
                                 bt_major=True)
            head_link.payload = d if head_link.payload is None else head_link.payload + d
            dl = _dummy_grad(logit)
        elif loss is not None:
            dl = H.mask_istft_bwd(None, logit, obs, wsyn, size, shift, fading, loss=loss)
        else:
            dl = H.mask_istft_bwd(dy, logit, obs, wsyn, size, shift, fading)
        return (dl, None, None, None, None, None, None, None, None, None)


def mask_istft(logit, obs, wsyn, N, size=1024, shift=256, fading=True, tgt=None):
    """logit [B,K,T,F], obs complex [B,T,F] -> time_estimate [B,K,N] (differentiable w.r.t. logit)."""
    if tgt is not None and tuple(tgt.shape) != (logit.shape[0], logit.shape[1], N):
        tgt = None
    fold = H.FOLD_TAIL and logit.requires_grad
    loss_link = _Link() if fold and tgt is not None and H.FOLD_TAIL != 3 else None
    head_link = getattr(logit, "_tssep_head_link", None) if fold and H.FOLD_TAIL != 2 else None
    out = _MaskISTFT.apply(logit, obs, wsyn, N, size, shift, fading, tgt, head_link, loss_link)
    y = out[0]
    if len(out) > 1:
        y._tssep_absdiff = (out[1], tgt.data_ptr(), tuple(tgt.shape))
        y._tssep_loss_link = loss_link
    return y


# ------------------------------------------------------------------------------ losses
def _fused_absdiff(est, tgt):
    """Partial sums the fused mask-head + iSTFT forward left on `est` for exactly this target."""
    info = getattr(est, "_tssep_absdiff", None)
    if info is not None and info[1] == tgt.data_ptr() and info[2] == tuple(tgt.shape):
        return info[0]
    return None


def _loss_link(est, part):
    """The hand-off to the fused tail that produced `est` (and `part` for this very target), if any."""
    return getattr(est, "_tssep_loss_link", None) if part is not None else None


class _LogMAE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, est, tgt, part=None, link=None):
        if part is not None:
            loss, sums = H.logmae_finalize(part, *est.shape)
        else:
            loss, sums = H.logmae_fwd(est, tgt)
        ctx.save_for_backward(est, tgt, sums)
        ctx.link = link
        return loss

    @staticmethod
    def backward(ctx, g):
        est, tgt, sums = ctx.saved_tensors
        if ctx.link is not None and ctx.link.payload is None:     # the tail's backward forms this gradient itself
            ctx.link.payload = (est, tgt, sums, g)
            return _dummy_grad(est), None, None, None
        return H.logmae_bwd(est.contiguous(), tgt.contiguous(), sums, g), None, None, None


def log_mae(est, tgt):
    part = _fused_absdiff(est, tgt)
    return _LogMAE.apply(est, tgt, part, _loss_link(est, part))


class _MAE(torch.autograd.Function):
    """sum_k mean_n |e - t| (loss.py:214-216): the argument of LogMAE's logarithm."""

    @staticmethod
    def forward(ctx, est, tgt, part=None, link=None):
        _, sums = H.logmae_finalize(part, *est.shape) if part is not None else H.logmae_fwd(est, tgt)
        ctx.save_for_backward(est, tgt)
        ctx.link = link
        return sums

    @staticmethod
    def backward(ctx, g):
        est, tgt = ctx.saved_tensors
        if ctx.link is not None and ctx.link.payload is None:
            ctx.link.payload = (est, tgt, None, g)
            return _dummy_grad(est), None, None, None
        return H.logmae_bwd(est.contiguous(), tgt.contiguous(), None, g), None, None, None


def mae(est, tgt):
    part = _fused_absdiff(est, tgt)
    return _MAE.apply(est, tgt, part, _loss_link(est, part))


class _VadBCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logit, vad):
        loss, xmean = H.vadbce_fwd(logit, vad)
        ctx.save_for_backward(xmean, vad)
        ctx.F = logit.shape[-1]
        return loss

    @staticmethod
    def backward(ctx, g):
        xmean, vad = ctx.saved_tensors
        return H.vadbce_bwd(xmean, vad, g, ctx.F), None


def vad_bce(logit, vad):
    return _VadBCE.apply(logit, vad)

"""The four model families of the native plans (zhusuan_amd/plans/dense.py
holds what they share: the packed state, the transition, the start
evaluation, the arithmetic choice, the C-side block runs).  A family says how
its likelihood kernel sees the state (`_layout`), lays the model's tensors out
for it (`_operands`), launches it (`_evaluate`) and fills its fields of the
zshmc_model_plan descriptor (`_describe`)."""
import math

import torch

from .. import _capi
from .base import _Unsupported
from .dense import _DenseLikelihoodPlan, _aligned16


class _LinearBernoulliPlan(_DenseLikelihoodPlan):
    """y ~ Bernoulli(w @ X^T [+ w2 @ X2^T ...] [+ b], group_ndims=1): one
    latent per term, up to 1 024 packed columns (univariate.py:398-403;
    csrc/linear_bernoulli*.hip, csrc/b3_kernel.h OP 0)."""
    kind = 'linear_bernoulli'
    takes_bf16x3 = True
    one_launch_capable = True

    def _layout(self, C, D, ld, f32):
        self.width, self.block = self._ops.likelihood_plan(ld)
        self.lik_rows = C
        return self.width != ld

    def _operands(self, inner, obs):
        ops = self._ops
        if self.packed:
            self.inner = _aligned16(ops.packed_design(
                inner, int(obs.shape[0]), self.device, self.width))
        else:
            self.inner = _aligned16(ops._padded_x(inner[0], self.width))
        self.obs = _aligned16(obs.detach().to(torch.float32).contiguous())
        return self.inner.shape[0]

    def _evaluate(self, w, q, grad, ll, ll_ptr, ws, stream):
        if self.inner_image is not None:
            _capi.call('zshmc_linear_bernoulli_log_lik_bf16x3', w.data_ptr(),
                       self.inner_image.data_ptr(), self.obs.data_ptr(),
                       self.n_chains, self.inner.shape[0], self.width,
                       ll_ptr, grad.data_ptr(), self.splits,
                       _capi.ptr(ws), stream)
            return
        _capi.call('zshmc_linear_bernoulli_log_lik', w.data_ptr(),
                   self.inner.data_ptr(), self.obs.data_ptr(),
                   self.n_chains, self.inner.shape[0], self.width,
                   ll_ptr, grad.data_ptr(), self.splits,
                   _capi.ptr(ws), stream)


class _MixtureMultinomialPlan(_DenseLikelihoodPlan):
    """x ~ UnnormalizedMultinomial(log(softmax(eta) @ phi),
    normalize_logits=False) (lntm_mcem.py:33-48, multivariate.py:435-443): one
    latent, up to 1 024 topics; rows are chain * n_docs + doc, one counts row
    per document (csrc/linear_bernoulli*.hip multinomial mode, csrc/b3_kernel.h
    OP 1 -- one document per workgroup, or packed rows)."""
    kind = 'mixture_multinomial'
    softmax = True
    takes_bf16x3 = True
    one_launch_capable = True
    obs_sp = None          # (compacted counts, rows, offsets) or None
    sparse_rows = False    # csrc/sparse_multinomial.hip runs the likelihood

    def _layout(self, C, D, ld, f32):
        self.width, self.block = self._ops.likelihood_plan(ld)
        self.lik_rows = C
        return True                        # theta = softmax(q)

    def _operands(self, inner, obs):
        ops = self._ops
        phi, x = inner[0], obs
        self._counts_src = x
        self.inner = _aligned16(ops._padded_phi_t(phi, self.width))
        # (rows padded to 32 floats: what the packed-rows form of the
        # bf16x3 kernel wants; the other kernels take any stride)
        self.obs, self.obs_stride = ops._padded_counts(x, 32)
        self.obs = _aligned16(self.obs)
        if self.n_chains % self.obs.shape[0] != 0:
            raise ValueError("counts rows do not divide the chain rows")
        return self.inner.shape[0]

    def _chains_per_counts_row(self):
        return self.n_chains // self.obs.shape[0]

    def _inner_rows_run(self, n_inner):
        # One document per workgroup on the bf16x3 kernel: the tile loop runs
        # over the document's OWN vocabulary (a bag of words is sparse, and a
        # word with a zero count contributes exactly nothing:
        # multivariate.py:435-443) where that is the smaller job
        ops = self._ops
        self.obs_sp = None
        self.sparse_rows = False
        self.n_inner_run = n_inner
        n_docs = self.obs.shape[0]
        if self.inner_image is None:
            # exact fp32: a SMALL problem (the reference's own E-step: one
            # chain x 100 documents) runs row by row over each row's own words
            # on the vector ALU (csrc/sparse_multinomial.hip) -- no tile
            # pipeline to fill, 8 % of the dense flops
            if self._sparse_rows_fit(n_inner, ops.SPARSE_ROWS_MAX):
                vals, rows, off, total = ops.counts_csr(self._counts_src)
                self.obs_sp = (vals, rows, off)
                self.sparse_rows = True
                self.traj_capacity = 0     # (not in the one-launch kernel)
                self.n_inner_run = max(32, total // n_docs)
            return self.n_inner_run
        if n_inner * self.width * 6 >= (1 << 31):
            return n_inner
        per_doc = self.n_chains // n_docs
        if self.packed_rows and per_doc < ops.BF16X3_SPARSE_MIN_CHAINS:
            return n_inner
        vals, rows, off, total = ops.counts_csr(self._counts_src)
        fill = total / float(n_docs * n_inner)
        # Chain axes that do not fill one-document workgroups (8 .. 127 chains
        # per document): a workgroup of `per_doc` valid chains over the
        # document's own words still beats 128 packed rows over ALL words when
        # the words are few enough -- tiles x 128 / per_doc against V tiles.
        if self.packed_rows and \
                fill * ops.BF16X3_CHAIN_BLOCK / per_doc > \
                ops.BF16X3_SPARSE_MAX_FILL:
            return n_inner
        if fill <= ops.BF16X3_SPARSE_MAX_FILL:
            self.packed_rows = False
            self.obs_sp = (vals, rows, off)
            self.n_inner_run = max(32, total // n_docs)
        return self.n_inner_run

    def _sparse_rows_fit(self, n_inner, max_rows):
        """Whether the row-by-row kernel takes this problem: <= 256 padded
        topics, at most max_rows rows, counts at most half full."""
        ops = self._ops
        if self.width > 256 or self.lik_rows > max_rows:
            return False
        total = ops.counts_csr(self._counts_src)[3]
        return total <= ops.SPARSE_ROWS_MAX_FILL * self.obs.shape[0] * n_inner

    def _auto_prefers_fp32(self, n_inner, per_doc, n_docs):
        ops = self._ops
        fills = n_docs == 1 or per_doc % ops.BF16X3_CHAIN_BLOCK == 0 or \
            per_doc >= 8 * ops.BF16X3_CHAIN_BLOCK
        # (from ~16 chains per document on, partly filled one-document
        # workgroups over the documents' own words beat it: 5 000 documents,
        # K = 128, gpurun r06: 16 chains 1.52 ms against 2.80, 32: 1.56 / 5.4)
        if not fills and per_doc < 2 * ops.BF16X3_SPARSE_MIN_CHAINS and \
                self._sparse_rows_fit(n_inner, ops.SPARSE_ROWS_AUTO_MAX):
            return ('%d rows with a word list each: the row-by-row fp32 '
                    'kernel (csrc/sparse_multinomial.hip) beats the '
                    'packed-rows bf16x3 form up to %d rows' % (
                        self.lik_rows, ops.SPARSE_ROWS_AUTO_MAX))
        return None

    def _choose_splits(self, R, n_inner, per_cu):
        if not self.sparse_rows:
            return super(_MixtureMultinomialPlan, self)._choose_splits(
                R, n_inner, per_cu)
        # a workgroup per (row, slice of its word list): about eight per CU,
        # slices of at least one 32-word tile; S * R a multiple of 4 (the step adds
        # the partials itself: csrc/hmc_model_run.hip steps_take_parts)
        cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        import os
        # (E-step, 100 rows x ~960 words, gpurun r06: 6 slices of 160 words
        # 0.418 ms per transition, 11 x 96 0.334, 21 x 64 0.315, 30 x 32 0.329;
        # the MFMA kernel 0.509 -- profiles/r06q_*; env knobs for A/B)
        per_cu = float(os.environ.get('ZSHMC_SPARSE_WG_PER_CU', '8'))
        min_words = int(os.environ.get('ZSHMC_SPARSE_MIN_WORDS', '32'))
        want = max(1, min(int(per_cu * cus + R - 1) // R,
                          n_inner // min_words, 256))
        for s in range(want, min(want + 4, 257)):
            if (s * R) % 4 == 0:
                return s
        return want

    def _evaluate(self, w, q, grad, ll, ll_ptr, ws, stream):
        if self.sparse_rows:
            vals, rows, off = self.obs_sp
            _capi.call('zshmc_sparse_multinomial_log_lik', w.data_ptr(),
                       self.inner.data_ptr(), vals.data_ptr(),
                       rows.data_ptr(), off.data_ptr(), self.obs.shape[0],
                       self.n_chains, self.inner.shape[0], self.width, ll_ptr,
                       grad.data_ptr(), self.splits, _capi.ptr(ws), stream)
            return
        if self.inner_image is not None and self.obs_sp is not None:
            vals, rows, off = self.obs_sp
            _capi.call('zshmc_linear_multinomial_log_lik_bf16x3_sparse',
                       w.data_ptr(), self.inner_image.data_ptr(),
                       vals.data_ptr(), rows.data_ptr(), off.data_ptr(),
                       self.obs.shape[0], self.n_chains, self.inner.shape[0],
                       self.width, ll_ptr, grad.data_ptr(), self.splits,
                       _capi.ptr(ws), stream)
            return
        name = 'zshmc_linear_multinomial_log_lik'
        inner = self.inner
        if self.inner_image is not None:
            name, inner = name + '_bf16x3', self.inner_image
        _capi.call(name, w.data_ptr(), inner.data_ptr(), self.obs.data_ptr(),
                   self.obs.shape[0], self.obs_stride, self.n_chains,
                   self.inner.shape[0], self.width, ll_ptr, grad.data_ptr(),
                   self.splits, _capi.ptr(ws), stream)

    def _describe(self, d):
        super(_MixtureMultinomialPlan, self)._describe(d)
        d.obs_rows, d.obs_stride = self.obs.shape[0], self.obs_stride
        if self.obs_sp is not None:
            # (with inner_image: the bf16x3 kernel over the documents' own
            # vocabularies; without: the row-by-row vector-ALU form)
            vals, rows, off = self.obs_sp
            d.obs_sp_counts, d.obs_sp_rows = vals.data_ptr(), rows.data_ptr()
            d.obs_sp_off = off.data_ptr()


class _LinearCategoricalPlan(_DenseLikelihoodPlan):
    """y ~ Categorical(X @ w[c]^T) (univariate.py:496-548): w[c, 0:K, 0:F], K
    class rows of F features per chain; the likelihood kernel's "chain rows"
    are the (chain, class) pairs, `stride` of them per chain (K rounded up to
    a power of two; csrc/lb_ops.h, csrc/b3_kernel.h OP 2,
    csrc/hmc_model_seg.hip)."""
    kind = 'linear_categorical'
    segmented = True
    takes_bf16x3 = True

    def _layout(self, C, D, ld, f32):
        K, F = (int(v) for v in self.q[0].shape[-2:])
        self.n_classes, self.seg_len = K, F
        self.stride = self._ops.class_stride(K)
        self.width, self.block = self._ops.likelihood_plan(F, self.stride)
        self._block32 = self.block
        self.lik_rows = C * self.stride
        self.seg_ws = torch.empty(
            int(_capi.load().zshmc_model_seg_workspace(C, D)), **f32)
        return not (K == self.stride and F == self.width)

    def _operands(self, inner, obs):
        ops = self._ops
        self.inner = _aligned16(ops._padded_x(inner[0], self.width))
        self.obs = _aligned16(ops.labels_as_float(obs, self.n_classes))
        return self.inner.shape[0]

    def _fp32_block(self):
        return self._block32               # (depends on the class stride)

    def _resident_per_cu(self):
        if self.inner_image is None:
            return 1
        return super(_LinearCategoricalPlan, self)._resident_per_cu()

    def _evaluate(self, w, q, grad, ll, ll_ptr, ws, stream):
        name = 'zshmc_linear_categorical_log_lik'
        inner = self.inner
        if self.inner_image is not None:
            name, inner = name + '_bf16x3', self.inner_image
        _capi.call(name, w.data_ptr(), inner.data_ptr(), self.obs.data_ptr(),
                   self.lik_rows, self.inner.shape[0], self.width,
                   self.n_classes, self.stride, ll_ptr, grad.data_ptr(),
                   self.splits, _capi.ptr(ws), stream)

    def _describe(self, d):
        super(_LinearCategoricalPlan, self)._describe(d)
        d.seg_len, d.groups = self.seg_len, self.stride
        d.seg_ws = self.seg_ws.data_ptr()


class _GatheredDotPlan(_DenseLikelihoodPlan):
    """r_e ~ Normal(sigmoid(u[i_e] . v[j_e]), std) over a pair list
    (pmf_hmc.py:19-31): the latent is one of the two factor tables,
    [chains, n, D] -- a handful of chains of 10^4..10^5 elements.  The
    gradient comes back as a plain [C, n * D] matrix: one "segment" per chain
    (csrc/gather_dot.hip, csrc/hmc_model_seg.hip)."""
    kind = 'gathered_dot'
    segmented = True

    def _layout(self, C, D, ld, f32):
        self.n_classes, self.seg_len, self.stride = 1, D, 1
        self.width = ld
        self.lik_rows = C
        self.seg_ws = torch.empty(
            int(_capi.load().zshmc_model_seg_workspace(C, D)), **f32)
        self.lp_const = torch.zeros(C, **f32)
        self._host_scalars = {}
        self._logstd_dev = torch.zeros(8, **f32)
        return False

    def _operands(self, inner, obs):
        self._refresh_gathered_dot(inner, obs)
        return None

    def _host_scalar(self, t):
        """float(t) of a one-element device tensor, read once per (storage,
        version): the per-run path does not synchronise."""
        key = (t.data_ptr(), t._version)
        hit = self._host_scalars.get(key)
        if hit is None:
            if len(self._host_scalars) > 64:
                self._host_scalars.clear()
            hit = self._host_scalars[key] = (float(t.item()), t)
        return hit[0]

    def _refresh_gathered_dot(self, inner, obs):
        """inner = [side ('u' | 'v': which table the latent is), other table,
        select (latent side), select (other side) or None, likelihood spread ('std' | 'logstd', tensor), constant nodes
        [(observed tensor, mean, (how, spread))...]]."""
        ops = self._ops
        self.side, other, sel_lat, sel_other, spread, consts = inner
        self.splits = 1
        q = self.q[0]
        n_lat, D = int(q.shape[-2]), int(q.shape[-1])
        self.n_lat, self.n_dim = n_lat, D
        self.other = _aligned16(other.detach().to(torch.float32).contiguous())
        self.n_other = int(self.other.shape[-2])
        E = int(sel_lat.numel())
        self.n_pairs = E
        # CSR view of the pair list by the latent's rows (deterministic
        # scatter of the gradient) -- cached per index tensor version
        self.idx_lat, self.seg, self.order = ops._pair_csr(
            sel_lat, n_lat, 'native_lat')
        if sel_other is None:       # `other` is already gathered pair by pair
            if getattr(self, '_iota', None) is None or \
                    self._iota.numel() != E:
                self._iota = torch.arange(E, dtype=torch.int32,
                                          device=self.device)
            self.idx_other = self._iota
        else:
            self.idx_other = ops._pair_csr(sel_other, self.n_other,
                                           'native_other')[0]
        r = obs.detach().to(torch.float32).contiguous()
        if r.numel() == E:
            self.obs, self.obs_rows = r.reshape(-1), 1
        elif r.numel() == self.n_chains * E:
            self.obs, self.obs_rows = r.reshape(-1), self.n_chains
        else:
            raise ValueError("HMC (native gathered_dot plan): %d observed "
                             "ratings for %d pairs" % (r.numel(), E))
        how, sp = spread
        sp_v = self._host_scalar(sp)
        self.lik_logstd = math.log(sp_v) if how == 'std' else sp_v
        need = int(_capi.load().zshmc_gather_dot_normal_workspace(
            self.n_chains, E))
        # likelihood + gradient in one pass over the pair list where the rows
        # are <= 128 floats, a multiple of 4 (csrc/gather_dot.hip:
        # gd_fused_kernel): the CSR view cut into segments, the other side's
        # indices and the ratings in CSR order
        self.gd_fused = D % 4 == 0 and D <= 128 and E > 0
        if self.gd_fused:
            key = (self.seg.data_ptr(), self.order.data_ptr(),
                   self.idx_other.data_ptr(), self.idx_other._version)
            if getattr(self, '_gd_seg_key', None) != key:
                self._gd_seg = ops._csr_segments(self.seg, E)
                self._gd_idx_csr = self.idx_other[self.order.long()].contiguous()
                self._gd_seg_key = key
            self._gd_obs_csr = _aligned16(self.obs.view(
                self.obs_rows, E)[:, self.order.long()].contiguous())
            n_seg = int(self._gd_seg[1].numel())
            # (partial sums rounded up to 4 floats: the per-segment gradient
            # rows behind them are written with 16-byte stores)
            groups = self.n_chains * n_seg
            need = max(need, groups * D + (groups + 3) // 4 * 4)
        if self._ws is None or self._ws.numel() < max(need, 1):
            self._ws = torch.empty(max(need, 1), dtype=torch.float32,
                                   device=self.device)
        if getattr(self, 'g_pairs', None) is None or \
                self.g_pairs.numel() < self.n_chains * max(E, 1):
            self.g_pairs = torch.empty(self.n_chains * max(E, 1),
                                       dtype=torch.float32, device=self.device)
        # the observed nodes that do not depend on the latent: their
        # log-densities (a constant of this run) join every log-joint value
        stream = _capi.current_stream()
        if len(consts) > 1:
            raise _Unsupported('more than one constant node in the joint')
        if consts:
            x, mean, (chow, csp) = consts[0]
            xs = _aligned16(x.detach().to(torch.float32).contiguous())
            cols = xs.numel() // self.n_chains
            cv = self._host_scalar(csp)
            _capi.call('zshmc_state_set', self._logstd_dev.data_ptr(), 0,
                       math.log(cv) if chow == 'std' else cv, stream)
            data_shape = tuple(xs.shape[len(self.chain_shape):])
            m = mean.detach().to(torch.float32)
            if m.numel() == 1:
                m, mode = m.reshape(1), _capi.BCAST_SCALAR
            elif tuple(m.shape[-len(data_shape):]) == data_shape and \
                    m.numel() == cols:
                m, mode = _aligned16(m.contiguous().reshape(-1)), \
                    _capi.BCAST_ROW
            else:
                m, mode = _aligned16(m.expand(xs.shape).contiguous()), \
                    _capi.BCAST_FULL
            self._const_keep = (xs, m)
            _capi.call('zshmc_normal_log_prob', xs.data_ptr(), m.data_ptr(),
                       self._logstd_dev.data_ptr(), self.lp_const.data_ptr(),
                       self.n_chains, cols, mode, _capi.BCAST_SCALAR, 1,
                       stream)
        else:
            _capi.call('zshmc_zero', self.lp_const.data_ptr(),
                       4 * self.n_chains, stream)

    def _evaluate(self, w, q, grad, ll, ll_ptr, ws, stream):
        # rating terms + d/d logit in one pass over the pairs, then the
        # deterministic scatter into the latent's rows
        lat_is_u = self.side == 'u'
        if self.gd_fused:
            sp, sr, sf, lr = self._gd_seg
            _capi.call(
                'zshmc_gather_dot_normal_lik_grad', q.data_ptr(),
                self.other.data_ptr(), sp.data_ptr(), sr.data_ptr(),
                sf.data_ptr(), lr.data_ptr() if lr.numel() else None,
                lr.numel(), self._gd_idx_csr.data_ptr(),
                self._gd_obs_csr.data_ptr(), self.obs_rows,
                self.lik_logstd, self.lp_const.data_ptr(), self.n_chains,
                self.n_lat, self.n_other, self.n_pairs, sr.numel(),
                self.n_dim, grad.data_ptr(), ll.data_ptr(),
                self._ws.data_ptr(), stream)
            return
        _capi.call(
            'zshmc_gather_dot_normal_lik',
            q.data_ptr() if lat_is_u else self.other.data_ptr(),
            self.other.data_ptr() if lat_is_u else q.data_ptr(),
            (self.idx_lat if lat_is_u else self.idx_other).data_ptr(),
            (self.idx_other if lat_is_u else self.idx_lat).data_ptr(),
            self.obs.data_ptr(), self.obs_rows, self.lik_logstd,
            self.lp_const.data_ptr(), self.n_chains,
            self.n_lat if lat_is_u else self.n_other,
            self.n_other if lat_is_u else self.n_lat, self.n_pairs,
            self.n_dim, self.g_pairs.data_ptr(), ll.data_ptr(),
            self._ws.data_ptr(), stream)
        if self.n_pairs:
            _capi.call('zshmc_gather_dot_grad', self.other.data_ptr(),
                       self.g_pairs.data_ptr(), self.seg.data_ptr(),
                       self.order.data_ptr(), self.idx_other.data_ptr(),
                       self.n_chains, self.n_lat, self.n_other,
                       self.n_pairs, self.n_dim, grad.data_ptr(),
                       stream)
        else:
            _capi.call('zshmc_zero', grad.data_ptr(),
                       4 * grad.numel(), stream)

    def _describe(self, d):
        d.seg_len, d.groups = self.seg_len, self.stride
        d.seg_ws = self.seg_ws.data_ptr()
        d.inner, d.n_inner = self.other.data_ptr(), self.n_other
        d.obs, d.obs_rows = self.obs.data_ptr(), self.obs_rows
        d.gd_latent_is_u = int(self.side == 'u')
        d.gd_idx_latent = self.idx_lat.data_ptr()
        d.gd_idx_other = self.idx_other.data_ptr()
        d.gd_seg, d.gd_order = self.seg.data_ptr(), self.order.data_ptr()
        d.gd_n_latent, d.gd_n_pairs = self.n_lat, self.n_pairs
        d.gd_n_dim, d.gd_logstd = self.n_dim, self.lik_logstd
        d.gd_lp_const = self.lp_const.data_ptr()
        d.gd_g_pairs = self.g_pairs.data_ptr()
        if self.gd_fused:
            sp, sr, sf, lr = self._gd_seg
            d.gd_seg_ptr, d.gd_seg_row = sp.data_ptr(), sr.data_ptr()
            d.gd_seg_first = sf.data_ptr()
            d.gd_long_rows = lr.data_ptr() if lr.numel() else None
            d.gd_n_seg, d.gd_n_long = sr.numel(), lr.numel()
            d.gd_idx_other_csr = self._gd_idx_csr.data_ptr()
            d.gd_obs_csr = self._gd_obs_csr.data_ptr()


FAMILIES = {cls.kind: cls for cls in (
    _LinearBernoulliPlan, _MixtureMultinomialPlan, _LinearCategoricalPlan,
    _GatheredDotPlan)}

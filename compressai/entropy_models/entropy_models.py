"""Entropy models of the MASIC codec, HIP-backed (reference
compressai/entropy_models/entropy_models.py:56-866).

On the hot path: `EntropyBottleneck.forward/loss`, `_quantize`, and
`GaussianMixtureConditional(_gf).forward`.  Bitstream side: `EntropyBottleneck.update / compress /
decompress` and the generic `EntropyModel.compress / decompress` run on the host-side rANS coder of libmasic_hip.so
(masic_amd/csrc/rans.hip), byte-exact with the reference's `compressai.ans`; symbols come from the HIP `symbols` kernel.
The Gaussian-conditional tables (`update_scale_table`) belong to HSIC.compress, row 8(f)-1: next.
Parameter / buffer names are the reference's, so state dicts interchange (248 tensors for HSIC).
"""
import numpy as np
import torch
import torch.nn as nn

from compressai.ops import LowerBound
from masic_amd import ops as _hip
from masic_amd.fresh import stamp as _stamp

from masic_amd import rans as _rans

_NEXT = "Gaussian-conditional coding tables belong to HSIC.compress / decompress, the 'next' row 8(f)-1 of SURVEY.md; not built yet"


def pmf_to_quantized_cdf(pmf, precision=16):
    """reference entropy_models.py:50-53"""
    return torch.from_numpy(_rans.pmf_to_quantized_cdf(pmf.detach().cpu().numpy(), precision).astype(np.int32))


class EntropyModel(nn.Module):
    def __init__(self, likelihood_bound=1e-9, entropy_coder=None, entropy_coder_precision=16):
        super().__init__()
        from compressai import available_entropy_coders, get_entropy_coder
        coder = get_entropy_coder() if entropy_coder is None else entropy_coder
        if coder not in available_entropy_coders():
            raise ValueError(f'Unknown entropy coder "{coder}" (available: {", ".join(available_entropy_coders())})')
        self.entropy_coder_name = coder
        self.entropy_coder_precision = int(entropy_coder_precision)
        self.likelihood_bound = float(likelihood_bound)
        self.use_likelihood_bound = likelihood_bound > 0
        if self.use_likelihood_bound:
            self.likelihood_lower_bound = LowerBound(likelihood_bound)
        # filled by update()
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())

    def forward(self, *args):
        raise NotImplementedError()

    def _get_noise_cached(self, x):
        """U(-1/2, 1/2) of x's shape from the device's global generator, one draw per call in call order (reference :89-96: one cached
        buffer per module, refilled by every call -- the draw order of SURVEY.md appendix D is what parity depends on, not the buffer).
        A FRESH tensor per call: gaussian1 serves three draws of a training forward, and with the entropy chains on side streams
        (HSIC._forward_graph) a refill of a shared buffer on one stream would race the previous draw's reader on another."""
        return x.new_empty(x.size()).uniform_(-0.5, 0.5)

    def _quantize(self, inputs, mode, means=None):
        if mode not in ("noise", "dequantize", "symbols"):
            raise ValueError(f'Invalid quantization mode: "{mode}"')
        if mode == "noise":
            noise = self._get_noise_cached(inputs)
            if torch.is_grad_enabled() and inputs.requires_grad:
                from masic_amd.autograd import AddNoiseFn
                return AddNoiseFn.apply(inputs, noise)
            return _hip.quantize(inputs.contiguous(), "noise", noise=noise)
        if mode == "symbols":
            med = None if means is None else means.reshape(-1).contiguous()
            if med is not None and med.numel() != inputs.shape[1]:
                raise NotImplementedError("per-element means in 'symbols' mode are outside the MASIC path")
            return _hip.symbols(inputs.contiguous(), med)
        if means is not None:
            raise NotImplementedError("'dequantize' with means is only reached through EntropyBottleneck.forward (fused)")
        return _hip.quantize(inputs.contiguous(), "dequantize")

    @staticmethod
    def _dequantize(inputs, means=None):
        if means is not None:
            return inputs.type_as(means) + means
        return inputs.float()

    def _pmf_to_cdf(self, pmf, tail_mass, pmf_length, max_length):
        """reference :136-142"""
        cdf = torch.zeros((len(pmf_length), int(max_length) + 2), dtype=torch.int32)
        for i, p in enumerate(pmf):
            prob = torch.cat((p[:pmf_length[i]], tail_mass[i]), dim=0)
            _cdf = pmf_to_quantized_cdf(prob, self.entropy_coder_precision)
            cdf[i, :_cdf.size(0)] = _cdf
        return cdf

    def _check_tables(self):
        """reference :144-163"""
        if self._quantized_cdf.numel() == 0:
            raise ValueError("Uninitialized CDFs. Run update() first")
        if self._quantized_cdf.dim() != 2:
            raise ValueError(f"Invalid CDF size {self._quantized_cdf.size()}")
        if self._offset.numel() == 0:
            raise ValueError("Uninitialized offsets. Run update() first")
        if self._offset.dim() != 1:
            raise ValueError(f"Invalid offsets size {self._offset.size()}")
        if self._cdf_length.numel() == 0:
            raise ValueError("Uninitialized CDF lengths. Run update() first")
        if self._cdf_length.dim() != 1:
            raise ValueError(f"Invalid offsets size {self._cdf_length.size()}")

    def _host_tables(self):
        return (self._quantized_cdf.cpu().numpy().astype(np.int32), self._cdf_length.cpu().numpy().astype(np.int32),
                self._offset.cpu().numpy().astype(np.int32))

    def compress(self, inputs, indexes, means=None):
        """Symbols on the device (HIP kernel), one rANS stream per batch element on the host (reference :165-196)."""
        if inputs.dim() != 4:
            raise ValueError("Invalid `inputs` size. Expected a 4-D tensor.")
        if inputs.size() != indexes.size():
            raise ValueError("`inputs` and `indexes` should have the same size.")
        self._check_tables()
        symbols = self._quantize(inputs, "symbols", means).cpu().numpy()
        idx = indexes.cpu().numpy().astype(np.int32)
        tables = self._host_tables()
        return [_rans.encode_with_indexes(symbols[i], idx[i], *tables) for i in range(symbols.shape[0])]

    def decompress(self, strings, indexes, means=None):
        """reference :199-239"""
        if not isinstance(strings, (tuple, list)):
            raise ValueError("Invalid `strings` parameter type.")
        if not len(strings) == indexes.size(0):
            raise ValueError("Invalid strings or indexes parameters")
        if indexes.dim() != 4:
            raise ValueError("Invalid `indexes` size. Expected a 4-D tensor.")
        self._check_tables()
        if means is not None:
            if means.size()[:-2] != indexes.size()[:-2]:
                raise ValueError("Invalid means or indexes parameters")
            if means.size() != indexes.size() and (means.size(2) != 1 or means.size(3) != 1):
                raise ValueError("Invalid means parameters")
        idx = indexes.cpu().numpy().astype(np.int32)
        tables = self._host_tables()
        values = np.stack([_rans.decode_with_indexes(s, idx[i], *tables).reshape(idx[i].shape) for i, s in enumerate(strings)])
        outputs = torch.from_numpy(values).to(self._quantized_cdf.device)
        return self._dequantize(outputs, means)


class EntropyBottleneck(EntropyModel):
    """Factorized prior of Balle et al. 2018 with per-channel 1-3-3-3-3-1 cumulative nets."""

    def __init__(self, channels, *args, tail_mass=1e-9, init_scale=10, filters=(3, 3, 3, 3), **kwargs):
        super().__init__(*args, **kwargs)
        self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        self.init_scale = float(init_scale)
        self.tail_mass = float(tail_mass)

        self._biases = nn.ParameterList()
        self._factors = nn.ParameterList()
        self._matrices = nn.ParameterList()
        dims = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1 / (len(self.filters) + 1))
        for i in range(len(self.filters) + 1):
            fill = float(np.log(np.expm1(1 / scale / dims[i + 1])))
            self._matrices.append(nn.Parameter(torch.full((self.channels, dims[i + 1], dims[i]), fill)))
            self._biases.append(nn.Parameter(torch.empty(self.channels, dims[i + 1], 1).uniform_(-0.5, 0.5)))
            if i < len(self.filters):
                self._factors.append(nn.Parameter(torch.zeros(self.channels, dims[i + 1], 1)))
        q = torch.Tensor([-self.init_scale, 0, self.init_scale])
        self.quantiles = nn.Parameter(q.repeat(self.channels, 1, 1))
        t = np.log(2 / self.tail_mass - 1)
        self.register_buffer("target", torch.Tensor([-t, 0, t]))

    def _medians(self):
        return self.quantiles[:, :, 1:2]

    def _table(self, differentiable=False):
        if differentiable:      # one autograd node: the table gradient goes back to the 14 parameters with one launch
            from masic_amd.autograd import eb_param_table
            table = eb_param_table(list(self._matrices), list(self._biases), list(self._factors))
            # the same values serve the detached uses until the parameters change (the auxiliary step of the same iteration)
            params = list(self._matrices) + list(self._biases) + list(self._factors)
            self.__dict__["_table_cache"] = (tuple((_stamp(p), p.data_ptr()) for p in params), table.detach())
            return table
        # inference: the [C, 58] table only changes with the parameters -- rebuilt per parameter version, not per forward
        params = list(self._matrices) + list(self._biases) + list(self._factors)
        key = tuple((_stamp(p), p.data_ptr()) for p in params)
        cache = self.__dict__.get("_table_cache")
        if cache is None or cache[0] != key:
            cache = (key, _hip.eb_param_table([m.detach() for m in self._matrices], [b.detach() for b in self._biases],
                                              [f.detach() for f in self._factors]))
            self.__dict__["_table_cache"] = cache
        return cache[1]

    def loss(self):
        """sum |logits(quantiles) - target| with all density parameters detached (reference :345-348)."""
        if torch.is_grad_enabled() and self.quantiles.requires_grad:
            from masic_amd.autograd import AuxLossFn
            return AuxLossFn.apply(self.quantiles, self._table(), self.tail_mass)
        return _hip.entropy_bottleneck_auxloss(self._table(), self.quantiles.detach().contiguous(), self.tail_mass)

    def forward(self, x):
        B, C, H, W = x.shape
        noise = None
        if self.training:
            # drawn in the reference's (C, 1, H*W*B) layout (reference :386-394)
            noise = self._get_noise_cached(x.new_empty((C, 1, H * W * B)))
        medians = self.quantiles.detach()[:, 0, 1].contiguous()
        lb = self.likelihood_bound if self.use_likelihood_bound else 0.0
        if self.training and torch.is_grad_enabled() and (x.requires_grad or self._matrices[0].requires_grad):
            from masic_amd.autograd import EntropyBottleneckFn
            return EntropyBottleneckFn.apply(x, self._table(differentiable=True), noise, medians, lb)
        return _hip.entropy_bottleneck(x.contiguous(), self._table(), medians, training=self.training, noise=noise,
                                       lik_bound=self.likelihood_bound if self.use_likelihood_bound else 0.0)

    @staticmethod
    def _build_indexes(size):
        N, C, H, W = size
        return torch.arange(C).view(1, -1, 1, 1).int().repeat(N, 1, H, W)

    def _logits_cumulative_host(self, inputs):
        """The cumulative nets on [C, 1, L] sample points (reference :350-371), evaluated with float32 torch ops on the host:
        `update` runs once per trained model, and byte-compatible streams need the reference's CPU arithmetic bit for bit."""
        logits = inputs
        for i in range(len(self.filters) + 1):
            logits = torch.matmul(torch.nn.functional.softplus(self._matrices[i].detach().cpu()), logits)
            logits = logits + self._biases[i].detach().cpu()
            if i < len(self._factors):
                logits = logits + torch.tanh(self._factors[i].detach().cpu()) * torch.tanh(logits)
        return logits

    def update(self, force=False):
        """Per-channel integer offsets and quantised CDF tables from the learned densities (reference :302-343)."""
        if self._offset.numel() > 0 and not force:
            return
        q = self.quantiles.detach().cpu()
        medians = q[:, 0, 1]
        minima = torch.clamp(torch.ceil(medians - q[:, 0, 0]).int(), min=0)
        maxima = torch.clamp(torch.ceil(q[:, 0, 2] - medians).int(), min=0)
        pmf_start = medians - minima
        pmf_length = maxima + minima + 1
        max_length = pmf_length.max()
        samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]
        lower = self._logits_cumulative_host(samples - 0.5)
        upper = self._logits_cumulative_host(samples + 0.5)
        sign = -torch.sign(lower + upper)
        pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
        tail_mass = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
        dev = self.quantiles.device
        self._offset = (-minima).to(dev)
        self._quantized_cdf = self._pmf_to_cdf(pmf, tail_mass, pmf_length, max_length).to(dev)
        self._cdf_length = (pmf_length + 2).to(dev)

    def compress(self, x):
        """reference :419-422"""
        indexes = self._build_indexes(x.size())
        medians = self._medians().detach().view(1, -1, 1, 1)
        return super().compress(x, indexes, medians)

    def decompress(self, strings, size):
        """reference :424-429"""
        output_size = (len(strings), self._quantized_cdf.size(0), size[0], size[1])
        indexes = self._build_indexes(output_size)
        medians = self._medians().detach().view(1, -1, 1, 1)
        return super().decompress(strings, indexes, medians)


class _GaussianBase(EntropyModel):
    def __init__(self, scale_table, *args, scale_bound=0.11, tail_mass=1e-9, **kwargs):
        super().__init__(*args, **kwargs)
        if not isinstance(scale_table, (type(None), list, tuple)):
            raise ValueError(f'Invalid type for scale_table "{type(scale_table)}"')
        if isinstance(scale_table, (list, tuple)) and len(scale_table) < 1:
            raise ValueError(f'Invalid scale_table length "{len(scale_table)}"')
        if scale_table and (scale_table != sorted(scale_table) or any(s <= 0 for s in scale_table)):
            raise ValueError(f'Invalid scale_table "({scale_table})"')
        self.register_buffer("scale_table", self._prepare_scale_table(scale_table) if scale_table else torch.Tensor())
        self.register_buffer("scale_bound", torch.Tensor([float(scale_bound)]) if scale_bound is not None else None)
        self.tail_mass = float(tail_mass)
        if scale_bound is None and scale_table:
            self._scale_bound_value = float(self.scale_table[0])
        elif scale_bound is not None and scale_bound > 0:
            self._scale_bound_value = float(scale_bound)
        else:
            raise ValueError("Invalid parameters")
        self.lower_bound_scale = LowerBound(self._scale_bound_value)

    @staticmethod
    def _prepare_scale_table(scale_table):
        return torch.Tensor(tuple(float(s) for s in scale_table))

    def _standardized_cumulative(self, inputs):
        return 0.5 * torch.erfc(float(-(2 ** -0.5)) * inputs)

    @staticmethod
    def _standardized_quantile(quantile):
        import scipy.stats
        return scipy.stats.norm.ppf(quantile)

    def update_scale_table(self, scale_table, force=False):
        """reference :494-501"""
        if self._offset.numel() > 0 and not force:
            return
        self.scale_table = self._prepare_scale_table(scale_table).to(self.scale_table.device)
        self.update()

    def update(self):
        """Quantised CDF tables of the zero-mean Gaussians of `scale_table` (reference :503-525), evaluated with float32 torch ops on
        the host for the same reason as EntropyBottleneck.update: byte-compatible streams need the reference's CPU arithmetic."""
        table = self.scale_table.detach().cpu()
        multiplier = -self._standardized_quantile(self.tail_mass / 2)
        pmf_center = torch.ceil(table * multiplier).int()
        pmf_length = 2 * pmf_center + 1
        max_length = torch.max(pmf_length).item()
        samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
        samples_scale = table.unsqueeze(1).float()
        upper = self._standardized_cumulative((0.5 - samples) / samples_scale)
        lower = self._standardized_cumulative((-0.5 - samples) / samples_scale)
        pmf = upper - lower
        tail_mass = 2 * lower[:, :1]
        dev = self.scale_table.device
        self._quantized_cdf = self._pmf_to_cdf(pmf, tail_mass, pmf_length, max_length).to(dev)
        self._offset = (-pmf_center).to(dev)
        self._cdf_length = (pmf_length + 2).to(dev)

    def build_indexes(self, scales):
        """index of the first scale_table entry >= LowerBound(scale) (reference :555-561); integer bookkeeping on the tensor's device"""
        scales = self.lower_bound_scale(scales)
        indexes = scales.new_full(scales.size(), len(self.scale_table) - 1).int()
        for s in self.scale_table[:-1]:
            indexes -= (scales <= s).int()
        return indexes


class GaussianConditional(_GaussianBase):
    """Single-Gaussian conditional (reference :433-562); not used by MASIC (it uses the mixture), provided for the API surface.
    forward = the K = 1 case of the mixture kernels: quantise about the means, likelihood of |y^ - mu| under N(0, LowerBound(sigma))."""

    def __init__(self, scale_table, *args, **kwargs):
        super().__init__(scale_table, *args, **kwargs)

    def _quantize_about(self, inputs, means):
        inputs = inputs.contiguous()
        if self.training:
            return _hip.quantize(inputs, "noise", noise=self._get_noise_cached(inputs))
        if means is None:
            return _hip.quantize(inputs, "dequantize")
        means = means.expand_as(inputs).contiguous()
        centred = _hip.elementwise(_hip.EW_AXPY, inputs, means, s0=1.0, s1=-1.0)     # x - mu
        return _hip.elementwise(_hip.EW_ADD, _hip.quantize(centred, "dequantize"), means)

    def _likelihood(self, inputs, scales, means=None):
        mu = torch.zeros_like(inputs) if means is None else means.expand_as(inputs).contiguous()
        _, lik = _hip.gmm_likelihood(inputs.contiguous(), scales.contiguous(), mu, torch.ones_like(inputs), 1, training=True,
                                     noise=torch.zeros_like(inputs), scale_bound=self._scale_bound_value, lik_bound=0.0)
        return lik

    def forward(self, inputs, scales, means=None):
        if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (inputs, scales, means)):
            raise NotImplementedError("GaussianConditional has no backward on the HIP path (MASIC trains the mixture, GaussianMixtureConditional_gf)")
        outputs = self._quantize_about(inputs, means)
        mu = torch.zeros_like(outputs) if means is None else means.expand_as(outputs).contiguous()
        lb = self.likelihood_bound if self.use_likelihood_bound else 0.0
        _, lik = _hip.gmm_likelihood(outputs, scales.contiguous(), mu, torch.ones_like(outputs), 1, training=True, noise=torch.zeros_like(outputs),
                                     scale_bound=self._scale_bound_value, lik_bound=lb)
        return outputs, lik


class GaussianMixtureConditional(_GaussianBase):
    """K-component Gaussian mixture conditional (reference :566-710); quantisation ignores the means
    (:695-697).  scales/means/weights carry K*M channels, component k of channel m at k*M + m."""

    def __init__(self, K, scale_table=None, mean_table=None, weight_table=None, *args, **kwargs):
        super().__init__(scale_table, *args, **kwargs)
        self.K = K

    def _likelihood(self, inputs, scales, means=None, weights=None):
        _, lik = _hip.gmm_likelihood(inputs.contiguous(), scales.contiguous(), means.contiguous(), weights.contiguous(),
                                     self.K, training=True, noise=torch.zeros_like(inputs),
                                     scale_bound=self._scale_bound_value, lik_bound=0.0)
        return lik

    def likelihood_of(self, y_hat, scales, means, weights, weights_are_logits=False):
        """Training mode, differentiable: the likelihood of an ALREADY quantised latent (y + noise drawn by the caller through
        `_quantize(y, "noise")`, at the position of this module's draw in the reference's order): forward() without its draw."""
        from masic_amd import autograd as A
        lb = self.likelihood_bound if self.use_likelihood_bound else 0.0
        return A.GmmLikFn.apply(y_hat, scales, means, weights, self.K, self._scale_bound_value, lb, bool(weights_are_logits))

    def forward(self, inputs, scales, means=None, weights=None, weights_are_logits=False):
        noise = self._get_noise_cached(inputs) if self.training else None
        lb = self.likelihood_bound if self.use_likelihood_bound else 0.0
        if self.training and torch.is_grad_enabled() and any(t.requires_grad for t in (inputs, scales, means, weights)):
            from masic_amd import autograd as A
            return A.GmmFn.apply(inputs, noise, scales, means, weights, self.K, self._scale_bound_value, lb, bool(weights_are_logits))
        return _hip.gmm_likelihood(inputs.contiguous(), scales.contiguous(), means.contiguous(), weights.contiguous(),
                                   self.K, training=self.training, noise=noise, weights_are_logits=weights_are_logits,
                                   scale_bound=self._scale_bound_value,
                                   lik_bound=self.likelihood_bound if self.use_likelihood_bound else 0.0)


class GaussianMixtureConditional_gf(GaussianMixtureConditional):
    """Pixel-wise ("gmm full") variant used by HSIC (reference :713-866; MASIC.py:658-659): same
    arithmetic, every latent position has its own K weights."""

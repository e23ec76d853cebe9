"""MLP construction with the reference's module layout + the flat parameter packing the kernels use.

`build_nn_from_config` yields the same nn.Sequential structure (hence the same state-dict keys
`0.weight, 0.bias, 2.weight, ...`, one shared activation instance, PReLU slope under `1.weight`) as
reference models/model_utils.py:4-39, so reference checkpoints load unchanged.  The arithmetic itself never
runs through these modules: `FlatParams` exposes their nn.Linear weights as views into ONE flat fp32 device
buffer in state-dict order, which is what the HIP kernels read and update in place.
"""
import torch
import torch.nn as nn

from .. import _lib

_ACTS = {"prelu": nn.PReLU, "relu": nn.ReLU, "leakyrelu": nn.LeakyReLU, "tanh": nn.Tanh, "identity": nn.Identity}


def build_nn_from_config(input_dim, output_dim, nn_config):
    hidden_size = nn_config['hidden_size']
    hidden_layer = nn_config['hidden_layer']
    activation_fn = nn_config['activation_fn']
    if activation_fn not in _ACTS:
        raise NotImplementedError('Unknown activation function: ' + str(activation_fn))
    # `use_layer_norm` (reference :22-29): ONE shared nn.LayerNorm after every hidden Linear but the first; lenv_mlp_forward
    # applies it (one-step API); the fused inner loops take plain MLPs only and their config builders say so
    norm = nn.LayerNorm(hidden_size) if nn_config.get("use_layer_norm", False) else nn.Identity()
    act_fn = _ACTS[activation_fn]()
    modules = [nn.Linear(input_dim, hidden_size), act_fn]
    for _ in range(hidden_layer - 1):
        modules += [nn.Linear(hidden_size, hidden_size), norm, act_fn]
    modules.append(nn.Linear(hidden_size, output_dim))
    return nn.Sequential(*modules)


def mlp_desc(net, activation_fn):
    """lenv_mlp_desc of an nn.Sequential built by build_nn_from_config."""
    linears = [m for m in net if isinstance(m, nn.Linear)]
    prelu = 0.25
    for m in net:
        if isinstance(m, nn.PReLU):
            prelu = float(m.weight.detach().reshape(-1)[0])
    return _lib.MlpDesc(linears[0].in_features, linears[0].out_features, len(linears) - 1, linears[-1].out_features,
                        _lib.ACT[activation_fn], prelu, 1 if any(isinstance(m, nn.LayerNorm) for m in net) else 0)


def _params_of(module, kinds):
    out = []
    for m in module.modules():                      # modules() yields a shared module once, at its first position
        if isinstance(m, kinds):
            out.append(m.weight)
            if m.bias is not None:
                out.append(m.bias)
    return out


def linear_params(module):
    """nn.Linear weights and biases in modules() order (the order GTN_worker.py:156-175 / GTN_master.py:281-296
    iterate in, == state-dict order): THE flat NES layout of theta / eps.  Only nn.Linear is perturbed and updated by the
    reference (GTN_worker.py:158,167; GTN_master.py:283,293): PReLU slopes and LayerNorm affines keep their initial values."""
    return _params_of(module, nn.Linear)


def mlp_params(module):
    """Every parameter lenv_mlp_forward reads for a build_nn_from_config net, in its flat order: the nn.Linear parameters
    plus -- for `use_layer_norm` nets -- the shared nn.LayerNorm's weight and bias at the module's (first) position."""
    return _params_of(module, (nn.Linear, nn.LayerNorm))


class FlatParams(object):
    """Re-homes every nn.Linear parameter of `module` as a view into one flat fp32 buffer."""

    def __init__(self, module, device):
        params = linear_params(module)
        n = sum(p.numel() for p in params)
        self.flat = torch.empty(n, dtype=torch.float32, device=device)
        off = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                self.flat[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + k].view(p.shape)
                off += k
        self.numel = n

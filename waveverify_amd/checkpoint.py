"""Checkpoint reading and writing: the reference's on-disk formats <-> {kind: state_dict} + NetConfigs.

Formats (all read with torch.load(..., weights_only=True): nothing in the file is executed):
  * atomic: one .pth dict {step, models{generator,detector,locator,discriminator}, optimizers,
    schedulers, tracker, config, ...} written by /root/reference/scripts/train.py:1589-1676;
    loader waveverify/core.py:324-426; directory search order best.pth, latest.pth, first
    (core.py:141-168).
  * legacy: <dir>/{generator,detector,locator}/model.pth (core.py:428-469).
State dicts may be in the stripped layout (plain `...weight`) or the live parametrized layout
(`...parametrizations.weight.original0/1`); both are accepted downstream (nets.HipNet).
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Mapping, Optional, Tuple

import torch

from .config import NetConfig, default_config

KINDS = ("generator", "detector", "locator")
_CLASS = {"generator": "Generator", "detector": "Detector", "locator": "Locator"}
_TAG = "parametrizations.weight.original"


def _load(path: Path):
    return torch.load(str(path), map_location="cpu", weights_only=True)


class UnsafeCheckpointError(RuntimeError):
    """A .pth file that torch.load(weights_only=True) refuses (it holds pickled objects beyond tensors
    and plain containers).  Never retried with weights_only=False."""


def _candidates(path: Path):
    files = sorted(path.glob("*.pth"))
    pref = [f for name in ("best.pth", "latest.pth") for f in files if f.name == name]
    return pref + [f for f in files if f not in pref]


def _scan_atomic(path: Path):
    """-> (file, checkpoint dict) of the first VALID atomic file in `path` (preference order best.pth,
    latest.pth, then the rest: core.py:141-168 for the order, core.py:343-356 for 'first file that loads
    and has a models dict'), or (None, None).  A file the safe loader refuses is remembered: if no
    valid file exists the refusal is raised instead of silently falling back to the legacy layout."""
    refused = None
    for f in _candidates(path):
        try:
            ck = _load(f)
        except Exception as e:                     # unreadable / refused by weights_only
            if refused is None:
                refused = (f, e)
            continue
        if isinstance(ck, dict) and "models" in ck:
            return f, ck
    if refused is not None:
        f, e = refused
        raise UnsafeCheckpointError(
            f"{f} cannot be read with torch.load(weights_only=True) ({type(e).__name__}: {str(e)[:200]}); "
            "waveverify_amd never unpickles arbitrary objects -- re-save the checkpoint with tensors and "
            "plain containers only (e.g. drop 'tracker' / argbind objects)")
    return None, None


def find_atomic_checkpoint_file(path: Path) -> Path:
    path = Path(path)
    if path.is_file() and path.suffix == ".pth":
        return path
    if not list(path.glob("*.pth")):
        raise FileNotFoundError(f"No atomic checkpoint files found in {path}")
    f, _ = _scan_atomic(path)
    if f is None:
        raise FileNotFoundError(f"No valid atomic checkpoint found in {path}")
    return f


def is_atomic_checkpoint(path: Path) -> bool:
    path = Path(path)
    if path.is_file() and path.suffix == ".pth":
        return True
    if path.is_dir():
        return _scan_atomic(path)[0] is not None
    return False


def _plain_key(sd: Mapping[str, object], key: str):
    """Shape of a possibly weight-normed tensor `key` ('...weight')."""
    if key in sd:
        return tuple(sd[key].shape)
    base = key[: -len("weight")]
    k1 = base + _TAG + "1"
    return tuple(sd[k1].shape) if k1 in sd else None


def infer_config(kind: str, sd: Mapping[str, object], base: Optional[NetConfig] = None) -> NetConfig:
    """Recover the architecture hyper-parameters from tensor shapes (the reference needs the
    argbind config for this; shapes carry the same information)."""
    cfg = base or default_config(kind)
    kw = cfg.to_dict()
    pre = _plain_key(sd, "encoder.conv_pre.1.conv.conv.weight")
    if pre is None:
        raise KeyError("state dict has no encoder.conv_pre weight")
    kw["channels_enc"], kw["kernel_size"] = pre[0], pre[2]
    ratios, s = [], 0
    while True:
        shp = _plain_key(sd, f"encoder.downsample.{s}.3.conv.conv.weight")
        if shp is None:
            break
        ratios.append(shp[2] // 2)
        s += 1
    kw["strides"] = list(reversed(ratios))                      # encoder walks them reversed
    n = 0
    while _plain_key(sd, f"encoder.blocks.0.{n}.block.1.conv.conv.weight") is not None:
        n += 1
    kw["n_residual_enc"] = n
    kw["residual_kernel_size"] = _plain_key(sd, "encoder.blocks.0.0.block.2.conv.conv.weight")[2]
    kw["n_fft_base"] = (_plain_key(sd, "encoder.spec_blocks.0.layer.conv.conv.weight")[1] - 1) * 2
    post = _plain_key(sd, "encoder.conv_post.2.conv.conv.weight")
    kw["dimension"] = post[0]
    kw["last_kernel_size"] = _plain_key(sd, "encoder.conv_post.1.conv.conv.weight")[2]
    kw["zero_init"] = "encoder.blocks.0.0.res_scale_param" in sd
    if "encoder.msg_embedding.0.weight" in sd:
        e = tuple(sd["encoder.msg_embedding.0.weight"].shape)
        kw["embedding_dim"], kw["msg_dimension"] = e[0], e[1]
        layers = 0
        while f"encoder.msg_embedding.{1 + 2 * layers}.weight" in sd:
            layers += 1
        kw["embedding_layers"] = layers
        b = 0
        while f"encoder.film_layers.0.{b}.gamma_layer.weight" in sd:
            b += 1
        kw["freq_bands"] = b or kw["freq_bands"]
    if kind == "generator":
        last = max(int(k.split(".")[2]) for k in sd if k.startswith("decoder.model."))
        kw["channels_dec"] = _plain_key(sd, f"decoder.model.{last}.conv.conv.weight")[1]
        n_groups = len(kw["strides"])
        kw["n_residual_dec"] = (last - 2 - 2) // n_groups - 4    # layout: seanet.py:1067-1204
    else:
        rev = tuple(sd["reverse_convolution.weight"].shape)
        kw["output_dim"] = rev[1]
        if kind == "detector":
            kw["nbits"] = tuple(sd["last_layer.weight"].shape)[0]
    return NetConfig(**kw)


def apply_argbind_config(kind: str, cfg: NetConfig, flat: Optional[Mapping[str, object]]) -> NetConfig:
    """Overlay an argbind-style flat dict ('Generator.res_scale_enc': ...) saved inside atomic
    checkpoints (train.py:1652) for the scalars shapes cannot tell."""
    if not flat:
        return cfg
    kw = cfg.to_dict()
    prefix = _CLASS[kind] + "."
    unsupported = {"norm": "weight_norm", "causal": True, "skip": "identity", "act_all": False,
                   "activation": "ELU", "spec": "stft", "spec_compression": "log",
                   "pad_mode": "constant", "inout_norm": True, "encoder_l2norm": True, "bias": True,
                   "expansion": 1, "groups": -1}
    for k, v in flat.items():
        if not isinstance(k, str) or not k.startswith(prefix):
            continue
        name = k[len(prefix):]
        if name in ("res_scale_enc", "res_scale_dec", "dilation_base"):
            if v is None:                          # Identity scale in the reference (seanet.py:1097-1108)
                raise NotImplementedError(f"{k}=None (no residual scaling) has no HIP path: only a "
                                          "numeric scale is supported")
            kw[name] = type(kw[name])(v)
        elif name in unsupported and v != unsupported[name]:
            raise NotImplementedError(
                f"{k}={v!r}: only the configuration the reference ships ({name}={unsupported[name]!r}) "
                "has a HIP path")
    return NetConfig(**kw)


def load_checkpoint(path) -> Tuple[Dict[str, dict], Dict[str, NetConfig]]:
    """-> ({kind: state_dict}, {kind: NetConfig}) for whichever of generator/detector/locator the
    checkpoint holds."""
    path = Path(path)
    if not path.exists():
        raise FileNotFoundError(f"Checkpoint not found: {path}")
    sds: Dict[str, dict] = {}
    flat = None
    if is_atomic_checkpoint(path):
        ck = _load(find_atomic_checkpoint_file(path))
        if not isinstance(ck, dict) or "models" not in ck:
            raise ValueError("Invalid atomic checkpoint format - missing 'models' key")
        flat = ck.get("config")
        for k in KINDS:
            if k in ck["models"]:
                sds[k] = dict(ck["models"][k])
    else:
        for k in KINDS:
            f = path / k / "model.pth"
            if f.exists():
                sds[k] = dict(_load(f))
    if not sds:
        raise FileNotFoundError(f"No generator/detector/locator weights found in {path}")
    cfgs = {k: apply_argbind_config(k, infer_config(k, sd), flat if isinstance(flat, dict) else None)
            for k, sd in sds.items()}
    return sds, cfgs


def stft_basis(n_fft: int) -> torch.Tensor:
    """The `...spec.weight` buffer of a CausalSTFT ([2F, 1, n_fft], F = n_fft/2 + 1), formed with the reference's own torch calls
    (/root/reference/modules/conv.py:1003-1020: float32 angle -2*pi/n_fft * k * n, cos rows then sin rows, periodic Hann window,
    norm "backward" = no scaling).  The nets recompute their basis at load; state dicts written here carry it so that the
    reference's `load_state_dict(strict=True)` finds every key."""
    import math
    n = torch.arange(n_fft, dtype=torch.float32).view(1, 1, n_fft)
    k = torch.arange(n_fft // 2 + 1, dtype=torch.float32).view(-1, 1, 1)
    ang = -2 * math.pi / n_fft * k * n
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=0) * torch.hann_window(n_fft, dtype=torch.float32)


def to_parametrized(sd: Mapping[str, object], cfg: NetConfig) -> Dict[str, torch.Tensor]:
    """A stripped state dict (plain `...weight`) in the live weight-norm layout a training step updates: what torch's weight_norm
    parametrization does when it is applied to an existing weight (`_WeightNorm.right_inverse`: original0 = ||w|| over all dimensions
    but the first, original1 = w), i.e. how the reference resumes from its own checkpoints.  Keys already parametrized pass through,
    and so do the DFT bases (`...spec.weight`: a learned parameter when the reference trained with spec_learnable: true -- the
    trainers run their forward on them and write them back)."""
    from .params import param_specs
    out: Dict[str, torch.Tensor] = {k: torch.as_tensor(v).float() for k, v in sd.items() if k.endswith("spec.weight")}
    for key, shape, role in param_specs(cfg):
        base = key[: -len("weight")] + _TAG if role == "wn" else None
        if role == "wn" and base + "0" in sd:
            out[base + "0"], out[base + "1"] = torch.as_tensor(sd[base + "0"]).float(), torch.as_tensor(sd[base + "1"]).float()
        elif role == "wn":
            w = torch.as_tensor(sd[key]).float()
            out[base + "0"] = w.reshape(w.shape[0], -1).norm(dim=1).reshape([w.shape[0]] + [1] * (w.dim() - 1))
            out[base + "1"] = w.clone()
        else:
            out[key] = torch.as_tensor(sd[key]).float()
    return out


_FIXED = {"activation": "ELU", "activation_kwargs": {"alpha": 1.0}, "norm": "weight_norm", "norm_kwargs": {}, "skip": "identity", "act_all": False,
          "expansion": 1, "groups": -1, "encoder_l2norm": True, "bias": True, "spec": "stft", "spec_compression": "log", "pad_mode": "constant",
          "causal": True, "inout_norm": True, "channels_audio": 1}


def argbind_config(cfgs: Mapping[str, NetConfig]) -> Dict[str, object]:
    """The flat 'Class.argument' dict the reference saves next to the weights (scripts/train.py:1652).  The reference's loader uses a
    present `config` INSTEAD of conf/base.yml and builds Generator / Detector / Locator under argbind.scope(config)
    (waveverify/core.py:226-236,272-276), so every constructor argument we leave out takes the CLASS default (zero_init=True,
    channels_enc=64, ...): the whole architecture is written, argument by argument (model/generator.py:63-104, detector.py:82-114,
    locator.py:84-115) -- the widths, strides, kernel sizes, residual scales, zero_init (which decides whether res_scale_param /
    scale_param exist at all), the heads' sizes, and the fixed options this library supports (apply_argbind_config's whitelist).
    tests/golden/state_dict_keys.json pins that the reference's constructors, given this dict, produce exactly our key set and shapes."""
    flat: Dict[str, object] = {}
    for kind, cfg in cfgs.items():
        c = _CLASS[kind]
        args = dict(sample_rate=int(cfg.sample_rate), dimension=int(cfg.dimension), channels_enc=int(cfg.channels_enc), n_fft_base=int(cfg.n_fft_base),
                    n_residual_enc=int(cfg.n_residual_enc), res_scale_enc=float(cfg.res_scale_enc), strides=[int(v) for v in cfg.strides],
                    kernel_size=int(cfg.kernel_size), last_kernel_size=int(cfg.last_kernel_size), residual_kernel_size=int(cfg.residual_kernel_size),
                    dilation_base=int(cfg.dilation_base), zero_init=bool(cfg.zero_init))
        args.update(_FIXED)
        if kind == "generator":
            args.update(msg_dimension=int(cfg.msg_dimension), channels_dec=int(cfg.channels_dec), n_residual_dec=int(cfg.n_residual_dec),
                        res_scale_dec=float(cfg.res_scale_dec), nbits=int(cfg.nbits), embedding_dim=int(cfg.embedding_dim),
                        embedding_layers=int(cfg.embedding_layers), freq_bands=int(cfg.freq_bands), final_activation="Tanh", spec_layer="1x1_zero",
                        spec_learnable=False)
        else:
            args.update(output_dim=int(cfg.output_dim))
            if kind == "detector":
                args.update(nbits=int(cfg.nbits))
        for k, v in args.items():
            flat[f"{c}.{k}"] = v
    return flat


def save_atomic_checkpoint(save_path, tag: str, models: Mapping[str, Mapping[str, torch.Tensor]], step: int = 0,
                           config: Optional[Mapping[str, object]] = None, extra: Optional[Mapping[str, object]] = None) -> Path:
    """Write <save_path>/<tag>.pth in the reference's atomic format (scripts/train.py:1589-1676): one dict {step, models{generator,
    detector, locator}, config, ...}, written to a temporary file and renamed into place.  Tensors and plain containers only, so
    the file loads with torch.load(weights_only=True) -- here and in the reference (waveverify/core.py:324-426)."""
    save_path = Path(save_path)
    save_path.mkdir(parents=True, exist_ok=True)
    final, tmp = save_path / f"{tag}.pth", save_path / f"{tag}.tmp"
    data: Dict[str, object] = {"step": int(step), "models": {k: dict(v) for k, v in models.items()}, "message_threshold": 0.5}
    if config is not None:
        data["config"] = dict(config)
    if extra:
        data.update(extra)
    try:
        torch.save(data, str(tmp))
        tmp.replace(final)
    except Exception:
        if tmp.exists():
            tmp.unlink()
        raise
    return final

#!/usr/bin/env python3
"""tests/golden/state_dict_keys.json: for a handful of configurations, the flat argbind dict THIS library writes into a checkpoint
(waveverify_amd.checkpoint.argbind_config) and the state-dict key -> shape table of the REFERENCE's Generator / Detector / Locator built
from exactly that dict -- i.e. what the reference's loader (waveverify/core.py:226-236,272-276: argbind.scope(checkpoint['config'])
around the three constructors) would construct for one of our files.  Only data is stored.  Build container only:

    python tests/golden/make_golden_keys.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from make_golden import _import_reference                       # noqa: E402
from waveverify_amd.checkpoint import _CLASS, argbind_config     # noqa: E402
from waveverify_amd.config import default_config                # noqa: E402

CASES = {
    "defaults": {},
    "zero_init_false": dict(zero_init=False),                   # what conf/base.yml ships
    "narrow": dict(channels_enc=16, dimension=32, n_fft_base=32),
    "narrow_zero_init_false_dilated": dict(channels_enc=16, dimension=32, zero_init=False, dilation_base=2, kernel_size=7, last_kernel_size=3),
}
GEN_ONLY = dict(narrow=dict(channels_dec=24, n_residual_dec=2, embedding_dim=32, freq_bands=2),
                narrow_zero_init_false_dilated=dict(channels_dec=24, n_residual_dec=1))


def main():
    torch, _, Generator, Detector, Locator = _import_reference()
    classes = {"generator": Generator, "detector": Detector, "locator": Locator}
    out = {}
    for name, kw in CASES.items():
        for kind, cls in classes.items():
            k2 = dict(kw)
            if kind == "generator":
                k2.update(GEN_ONLY.get(name, {}))
            if kind == "locator" and "dimension" in k2:
                k2["dimension"] = 16
            cfg = default_config(kind, **k2)
            flat = argbind_config({kind: cfg})
            prefix = _CLASS[kind] + "."
            net = cls(**{k[len(prefix):]: v for k, v in flat.items()})          # what argbind.scope(config) binds
            keys = {k: list(v.shape) for k, v in net.state_dict().items()}
            out[f"{name}/{kind}"] = dict(overrides=k2, config=flat, keys=keys)
            print(f"{name}/{kind}: {len(keys)} state-dict entries")
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()

"""Model configuration for the PGTFormer forward path.

`DEFAULT_NETWORK_G` restates the `network_g` block of the reference's test option file
(reference: options/release_test_stage_IIII_dont_need_align_version.yml:53-90) which is also what
`PGTFormer.from_pretrained` receives as constructor kwargs (reference: inference.py:109-121).
Keys the reference constructors swallow via **ignore_kwargs (stages_atten, window_size, num_head;
reference: archs/tdcrqvae3_arch.py:463,580) are accepted and ignored here as well.
"""
import copy
import json

DEFAULT_NETWORK_G = {
    "type": "PGTFormer",
    "w": 1,
    "adain": True,
    "checkpointing": False,
    "bottleneck_type": "rq",
    "embed_dim": 512,
    "n_embed": 1024,
    "latent_shape": [32, 32, 512],
    "code_shape": [32, 32, 1],
    "shared_codebook": True,
    "decay": 0.99,
    "restart_unused_codes": True,
    "loss_type": "mse",
    "latent_loss_weight": 0.25,
    "tf": 3,
    "ddconfig": {
        "double_z": False,
        "z_channels": 256,
        "resolution": 512,
        "in_channels": 3,
        "stages_atten": 4,
        "window_size": [5, 5, 5],
        "num_head": 8,
        "out_ch": 3,
        "ch": 64,
        "ch_mult": [1, 2, 4, 4, 8],
        "depths": [2, 2, 2, 2, 2],
        "num_heads": [8, 8, 8, 8, 8],
        "window_sizes": [[4, 4], [4, 4], [4, 4], [4, 4], [4, 4]],
        "num_frames": 3,
        "num_res_blocks": 1,
        "attn_resolutions": [32, 64, 128],
        "dropout": 0.0,
    },
}

# precision mode of `model.prepare(device, precision)` when none is named (archs/tdcrqvae3_arch.py: TDCRQVAE3.prepare)
DEFAULT_PRECISION = "x3f16"

# PGTFormer.__init__ defaults (reference: archs/pgtformer_arch.py:491-495)
PGTFORMER_DEFAULTS = {
    "dim_embd": 512,
    "n_head": 8,
    "n_layers": 9,
    "connect_list": ["32", "64", "128", "256"],
    "fix_modules": ["quantizer", "decoder", "conditionnet"],
    "w": 0,
    "detach_16": True,
    "adain": False,
    "tf": 3,
    "droprate": 0.0,
}


def default_config():
    """Return a deep copy of the shipping configuration (without the `type` key)."""
    cfg = copy.deepcopy(DEFAULT_NETWORK_G)
    cfg.pop("type", None)
    return cfg


def load_config(path):
    """Read a `network_g` config from a BasicSR YAML option file or an HF `config.json`."""
    if path.endswith((".yml", ".yaml")):
        import yaml

        with open(path, "r") as f:
            opt = yaml.safe_load(f)
        cfg = dict(opt["network_g"]) if "network_g" in opt else dict(opt)
    else:
        with open(path, "r") as f:
            cfg = json.load(f)
    cfg.pop("type", None)
    return cfg

"""Checkpoint surface of the reference: `PGTFormer.from_pretrained(dir)` = config.json -> constructor
kwargs, model.safetensors -> strict load (huggingface_hub's PyTorchModelHubMixin on the class, reference
archs/tdcrqvae3_arch.py:711, call site inference.py:118); `.pth` with params_ema / params; flat .safetensors.
CPU only: loading needs no GPU, running does."""
import json
import os

import pytest
import torch


@pytest.fixture(scope="module")
def saved(tmp_path_factory, cfg, full_sd):
    from pgtformer_amd import PGTFormer

    d = tmp_path_factory.mktemp("ckpt")
    m = PGTFormer(**cfg)
    m.load_state_dict(full_sd, strict=True)
    m.save_pretrained(str(d))
    return str(d), m


def test_from_pretrained_roundtrip(saved, manifest):
    from pgtformer_amd import PGTFormer

    d, m = saved
    assert sorted(os.listdir(d)) == ["config.json", "model.safetensors"]
    cfg = json.load(open(os.path.join(d, "config.json")))
    assert cfg["ddconfig"]["ch_mult"] == [1, 2, 4, 4, 8] and cfg["adain"] is True and cfg["n_embed"] == 1024
    m2 = PGTFormer.from_pretrained(d)                    # no device: constructed + strictly loaded, not prepared
    sd, sd2 = m.state_dict(), m2.state_dict()
    assert list(sd) == list(sd2) == list(manifest) and len(sd2) == 961
    for k in sd:
        assert torch.equal(sd[k], sd2[k]), k
    assert m2.enc_dt is None and not m2.training and not any(p.requires_grad for p in m2.parameters())
    assert m2.adain is True and m2.w == 1
    with pytest.raises(FileNotFoundError):
        PGTFormer.from_pretrained(os.path.join(d, "nope"))


def test_from_pretrained_is_strict(saved, tmp_path):
    from safetensors.torch import load_file, save_file
    from pgtformer_amd import PGTFormer

    d, _ = saved
    sd = load_file(os.path.join(d, "model.safetensors"))
    sd.pop("encoder.conv_in.bias")
    bad = tmp_path / "bad"
    bad.mkdir()
    save_file(sd, str(bad / "model.safetensors"))
    (bad / "config.json").write_text(open(os.path.join(d, "config.json")).read())
    with pytest.raises(RuntimeError):
        PGTFormer.from_pretrained(str(bad))


def test_load_architecture_formats(saved, tmp_path, monkeypatch):
    """directory / flat .safetensors / .pth{params_ema}: all reach the same strictly-loaded model (prepare() is stubbed:
    no GPU here)."""
    from safetensors.torch import load_file
    from pgtformer_amd import PGTFormer, driver

    d, m = saved
    monkeypatch.setattr(PGTFormer, "prepare", lambda self, device="cuda", precision="bf16x3": self)
    sd = load_file(os.path.join(d, "model.safetensors"))
    pth = tmp_path / "net_g.pth"
    torch.save({"params_ema": sd}, pth)
    for src in (d, os.path.join(d, "model.safetensors"), str(pth)):
        got = driver.load_architecture(weights=src).state_dict()
        assert all(torch.equal(got[k], v) for k, v in m.state_dict().items()), src
    # a pickle that is not a plain tensor dict is refused (weights_only=True)
    evil = tmp_path / "evil.pth"
    torch.save({"params": sd, "hook": print}, evil)
    with pytest.raises(Exception):
        driver.load_architecture(weights=str(evil))

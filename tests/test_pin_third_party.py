"""Skip-if-absent wrapper around tools/pin_third_party.py: on a box that has vocos / peft / torchaudio / vector_quantize_pytorch / zh_normalization
(this image has none of them: SURVEY 8c) every present package must agree with the repo's restatement of it -- rows V1 and P3 of the coverage table
lift the first time these run green."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("pin_third_party", os.path.join(ROOT, "tools", "pin_third_party.py"))
pin = importlib.util.module_from_spec(spec)
sys.modules["pin_third_party"] = pin
spec.loader.exec_module(pin)


@pytest.mark.parametrize("name", sorted(pin.SECTIONS))
def test_third_party_package_agrees_with_the_restatement(name):
    res = pin.SECTIONS[name]()
    if res["status"] == "absent":
        pytest.skip(f"{name} is not installed in this image")
    assert res["status"] == "pinned", res


def test_absent_checkpoints_are_reported_not_guessed(tmp_path):
    assert pin.pin_checkpoints(None, False)["status"] == "absent"
    assert pin.pin_checkpoints(str(tmp_path / "nowhere"), False)["status"] == "absent"


def test_synthetic_checkpoint_directory_passes_the_key_and_range_audit(tmp_path):
    """The audit itself is exercised on the synthetic checkpoint directory the end-to-end tests use (same key sets as the real files)."""
    from chatttsplus_amd import synth
    d = synth.write_checkpoints(str(tmp_path / "ckpt"), 1234)
    res = pin.pin_checkpoints(d, False)
    assert res["status"] == "pinned", res
    for name in ("GPT.pt", "Decoder.pt", "Vocos.pt", "DVAE_full.pt"):
        assert res[name]["missing_keys"] == [] and res[name]["fits_fp16_weights"], (name, res[name])

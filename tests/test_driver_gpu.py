"""The command-line driver end to end on the GPU: `python convnet.py ...` trains on dummy data, prints the
reference's per-batch lines, writes a checkpoint (pickle of {"model_state", "op"}), resumes from it with
-f, and --check-grads reports ALL TESTS PASSED (Python/convnet.py, Python/gpumodel.py:126-167, 267-294)."""
import os
import pickle
import re
import subprocess
import sys

import pytest

import modelutil

pytestmark = pytest.mark.gpu
PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-convnet_amd")


def run(args, timeout=300):
    env = dict(os.environ, CONVNET_SEED="7")
    r = subprocess.run([sys.executable, "-u", os.path.join(PKG, "convnet.py")] + args, capture_output=True, text=True, timeout=timeout, cwd=PKG, env=env)
    return r.returncode, r.stdout + r.stderr


def test_train_checkpoint_resume(tmp_path):
    cfg = ["--layer-def=" + os.path.join(modelutil.CONFIGS, "layers-80sec.cfg"), "--layer-params=" + os.path.join(modelutil.CONFIGS, "layer-params-80sec.cfg")]
    base = ["--data-path=" + str(tmp_path), "--save-path=" + str(tmp_path), "--train-range=1", "--test-range=1", "--data-provider=dummy-cn-3072", "--test-freq=2"]
    rc, out = run(base + cfg + ["--epochs=4"])
    assert rc == 0, out[-3000:]
    lines = re.findall(r"^(\d+)\.1\.\.\. logprob:  ([\d.]+), ([\d.]+) \(([\d.]+) sec\)", out, flags=re.M)
    assert len(lines) >= 2, out[-3000:]
    assert all(0.5 < float(l[1]) < 5 for l in lines)                # ~log(10) per case on random data
    assert "======================Test output======================" in out and "Saved checkpoint to" in out
    ckpts = [d for d in os.listdir(tmp_path) if d.startswith("ConvNet__")]
    assert len(ckpts) == 1
    files = sorted(os.listdir(tmp_path / ckpts[0]))
    assert files, "no checkpoint file"
    with open(tmp_path / ckpts[0] / files[-1], "rb") as f:
        dic = pickle.load(f)
    assert set(dic) == {"model_state", "op"} and "layers" in dic["model_state"]
    conv1 = [l for l in dic["model_state"]["layers"] if l["name"] == "conv1"][0]
    assert abs(conv1["weightsInc"][0]).max() > 0                     # syncWithHost wrote the trained state back

    rc, out = run(["-f", str(tmp_path / ckpts[0]), "--epochs=6"])
    assert rc == 0, out[-3000:]
    assert re.search(r"^[56]\.1\.\.\. logprob:", out, flags=re.M), out[-3000:]


def test_check_grads_flag():
    rc, out = run(["--check-grads=1", "--layer-def=" + os.path.join(modelutil.CONFIGS, "gc-small.def"),
                   "--layer-params=" + os.path.join(modelutil.CONFIGS, "gc-small.params"), "--data-provider=dummy-cn-432", "--mini=32"])
    assert rc == 0, out[-3000:]
    assert re.search(r"ALL \d+ TESTS PASSED", out), out[-3000:]


def test_resume_with_conv_to_local(tmp_path):
    """-f <checkpoint> --conv-to-local=conv2 (convnet.py:78-84 of the reference): the resumed run converts the
    layer, keeps training through the unshared kernels, and the next checkpoint holds the per-module filter bank."""
    cfg = ["--layer-def=" + os.path.join(modelutil.CONFIGS, "layers-80sec.cfg"), "--layer-params=" + os.path.join(modelutil.CONFIGS, "layer-params-80sec.cfg")]
    base = ["--data-path=" + str(tmp_path), "--save-path=" + str(tmp_path), "--train-range=1", "--test-range=1", "--data-provider=dummy-cn-3072", "--test-freq=1"]
    rc, out = run(base + cfg + ["--epochs=1"])
    assert rc == 0, out[-3000:]
    ckpt = tmp_path / [d for d in os.listdir(tmp_path) if d.startswith("ConvNet__")][0]
    rc, out = run(["-f", str(ckpt), "--conv-to-local=conv2", "--epochs=3"])
    assert rc == 0, out[-3000:]
    assert "Converted layer 'conv2' from convolutional to unshared, locally-connected" in out
    lines = re.findall(r"^([23])\.1\.\.\. logprob:  ([\d.]+)", out, flags=re.M)
    assert lines and all(0.5 < float(v) < 5 for _, v in lines), out[-3000:]
    with open(ckpt / sorted(os.listdir(ckpt), key=lambda s: [int(x) for x in s.split(".")])[-1], "rb") as f:
        dic = pickle.load(f)
    conv2 = [l for l in dic["model_state"]["layers"] if l["name"] == "conv2"][0]
    assert conv2["type"] == "local" and conv2["weights"][0].shape == (conv2["modules"] * 32 * 25, 32)
    bank = conv2["weights"][0].reshape(conv2["modules"], -1, 32)
    assert abs(bank[0] - bank[conv2["modules"] // 2]).max() > 0           # the modules have trained apart

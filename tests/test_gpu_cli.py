"""End-to-end smoke of the reference-compatible drivers on a real GPU (synthetic weights): argparse surface -> create_network ->
solver -> VAE decode -> uint8 -> files.  Small shapes; correctness of each stage is covered by the parity tests."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--image_size", "256", "--f", "8", "--num_in_channels", "4", "--num_out_channels", "4", "--random_weights", "--generator", "device",
          "--batch_size", "2", "--n_sample", "4"]


def _run(cmd, **kw):
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600, **kw)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_single_process_driver_euler_and_karras_heun(tmp_path):
    out = _run([sys.executable, "-m", "lfm_amd.test_flow_latent", "--model_type", "DiT-S/2", "--num_classes", "1", "--label_dropout", "0.",
                "--method", "euler", "--step_size", "0.25", "--save_dir", str(tmp_path / "a"), *COMMON])
    # default mode = one sample sheet named as the reference names it (test_flow_latent.py:288-298): 2 images of 256x256, nrow 8
    assert "Samples are save at" in out
    assert os.listdir(tmp_path / "a") == ["samples_cifar10_euler_1e-05_1e-05.jpg"]
    from PIL import Image

    assert Image.open(tmp_path / "a" / "samples_cifar10_euler_1e-05_1e-05.jpg").size == (512, 256)
    out = _run([sys.executable, "-m", "lfm_amd.test_flow_latent", "--model_type", "DiT-S/2", "--num_classes", "10", "--label_dropout", "0.1",
                "--cfg_scale", "1.5", "--use_karras_samplers", "--method", "heun", "--num_steps", "5", "--save_dir", str(tmp_path / "b"), *COMMON])
    assert "Samples are save at" in out and os.listdir(tmp_path / "b") == ["samples_cifar10_heun_5_cfg1.5.jpg"]
    # single-process --compute_fid: per-image JPEGs named by global index, written through the ROUNDING conversion
    out = _run([sys.executable, "-m", "lfm_amd.test_flow_latent", "--model_type", "DiT-S/2", "--num_classes", "1", "--label_dropout", "0.",
                "--method", "euler", "--step_size", "0.5", "--compute_fid", "--save_dir", str(tmp_path / "d"), *COMMON])
    assert sorted(os.listdir(tmp_path / "d"), key=lambda s: int(s.split(".")[0])) == ["0.jpg", "1.jpg", "2.jpg", "3.jpg"]


def test_batches_in_flight_write_the_same_files(tmp_path):
    """--compute_fid with two lanes (default) and with --in_flight 1: four batches of two images each; every JPEG must be byte-identical (the lanes are
    bit-identical to one lane, and the files are written in batch order under the same global indices)."""
    cmd = [sys.executable, "-m", "lfm_amd.test_flow_latent", "--model_type", "DiT-S/2", "--num_classes", "1", "--label_dropout", "0.", "--method", "euler",
           "--step_size", "0.25", "--compute_fid", *[("8" if COMMON[i - 1] == "--n_sample" else a) for i, a in enumerate(COMMON)]]
    _run(cmd + ["--save_dir", str(tmp_path / "two")])
    _run(cmd + ["--save_dir", str(tmp_path / "one"), "--in_flight", "1"])
    names = sorted(os.listdir(tmp_path / "two"), key=lambda s: int(s.split(".")[0]))
    assert names == [f"{i}.jpg" for i in range(8)] == sorted(os.listdir(tmp_path / "one"), key=lambda s: int(s.split(".")[0]))
    for nm in names:
        assert open(tmp_path / "two" / nm, "rb").read() == open(tmp_path / "one" / nm, "rb").read(), nm


def test_single_process_driver_dit_at_512(tmp_path):
    """--image_size 512 with a DiT-x/2: 64x64 latents = 1024 tokens per image through the graph-captured Euler solver, decoded at 512x512."""
    common = [("512" if a == "256" else a) for a in COMMON]
    out = _run([sys.executable, "-m", "lfm_amd.test_flow_latent", "--model_type", "DiT-S/2", "--num_classes", "1", "--label_dropout", "0.",
                "--method", "euler", "--step_size", "0.25", "--save_dir", str(tmp_path / "f"), *common])
    assert "Samples are save at" in out
    from PIL import Image

    assert Image.open(tmp_path / "f" / os.listdir(tmp_path / "f")[0]).size == (1024, 512)


def test_ddp_driver_one_rank_rccl(tmp_path):
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", "29517", "-m", "lfm_amd.test_flow_latent_ddp", "--model_type", "DiT-S/2", "--num_classes", "1",
                "--label_dropout", "0.", "--method", "euler", "--step_size", "0.5", "--compute_fid", "--save_dir", str(tmp_path / "c"), *COMMON])
    assert "sampled 4 images on 1 GPUs" in out
    assert sorted(os.listdir(tmp_path / "c"), key=lambda s: int(s.split(".")[0])) == ["0.jpg", "1.jpg", "2.jpg", "3.jpg"]


def test_launcher_with_a_reference_args_file(tmp_path):
    """python -m lfm_amd.run_test <test_args file>: the reference's run_test.sh expansion, solver overridden after '--' to keep it short."""
    f = tmp_path / "celeb256_dit.txt"
    f.write_text("MODEL_TYPE=DiT-L/2\nEPOCH_ID=475\nDATASET=celeba_256\nEXP=celeb_f8_dit\nMETHOD=dopri5\nSTEPS=0\nUSE_ORIGIN_ADM=false\nIMG_SIZE=256")
    out = _run([sys.executable, "-m", "lfm_amd.run_test", str(f), "--", "--random_weights", "--generator", "device", "--batch_size", "2",
                "--method", "euler", "--step_size", "0.5", "--save_dir", str(tmp_path / "e")])
    assert "Argument file:" in out and "Samples are save at" in out
    assert os.listdir(tmp_path / "e") == ["samples_celeba_256_euler_1e-05_1e-05.jpg"]

"""The reference's launcher scripts as one Python entry point (bash_scripts/run_test{,_cls,_ddp,_cls_ddp}.sh + test_args/*.txt).

    python -m lfm_amd.run_test test_args/celeb256_dit.txt                      # run_test.sh        (unconditional, one process)
    python -m lfm_amd.run_test test_args/imnet_dit.txt --cls                   # run_test_cls.sh    (ImageNet, CFG)
    python -m lfm_amd.run_test test_args/celeb256_dit.txt --ddp --num_gpus 8   # run_test_ddp.sh    (one process per GPU, --compute_fid)
    python -m lfm_amd.run_test test_args/imnet_dit.txt --cls --ddp             # run_test_cls_ddp.sh
    ... -- --random_weights --method euler --step_size 0.02                    # anything after "--" is appended to the flags

An argument file is a bash fragment of ``KEY=VALUE`` lines (values optionally double-quoted) with the keys MODEL_TYPE EPOCH_ID DATASET
EXP METHOD STEPS USE_ORIGIN_ADM IMG_SIZE CH_MULT ATTN_RES Bs BASE_CH CFG; the scripts ``source`` it, fill in defaults and expand it into
test_flow_latent{,_ddp}.py flags.  ``build_argv`` reproduces that expansion flag for flag (the lines the scripts keep commented out --
--measure_time / --use_karras_samplers / --compute_nfe -- are left to the caller)."""
import argparse
import os
import shlex
import subprocess
import sys

KEYS = ("MODEL_TYPE", "EPOCH_ID", "DATASET", "EXP", "METHOD", "STEPS", "USE_ORIGIN_ADM", "IMG_SIZE", "CH_MULT", "ATTN_RES", "Bs", "BASE_CH", "CFG")


def parse_args_file(text):
    """KEY=VALUE lines of a bash fragment -> dict of strings (what ``source`` would leave in the environment)."""
    out = {}
    for raw in text.splitlines():
        line = raw.strip()
        if not line or line.startswith("#"):
            continue
        if "=" not in line:
            raise ValueError(f"not a KEY=VALUE line: {raw!r}")
        key, val = line.split("=", 1)
        key = key.strip()
        parts = shlex.split(val, comments=True)
        out[key] = " ".join(parts)
    return out


def build_argv(cfg, cls=False, ddp=False):
    """The flag list run_test.sh / run_test_cls.sh / run_test_ddp.sh / run_test_cls_ddp.sh pass for this argument file."""
    v = dict(cfg)
    if cls:  # run_test_cls*.sh: fixed 256x256 ImageNet ADM/DiT geometry, Bs default 50
        v.setdefault("Bs", "50")
        argv = ["--exp", v["EXP"], "--dataset", v["DATASET"], "--batch_size", v["Bs"], "--epoch_id", v["EPOCH_ID"],
                "--image_size", "256", "--f", "8", "--num_in_channels", "4", "--num_out_channels", "4",
                "--nf", "256", "--ch_mult", "1", "2", "3", "4", "--attn_resolution", "16", "8", "4", "--num_res_blocks", "2",
                "--model_type", v["MODEL_TYPE"], "--num_classes", "1000", "--label_dim", "1000", "--label_dropout", "0.1",
                "--method", v["METHOD"], "--num_steps", v["STEPS"], "--cfg_scale", v.get("CFG", "")]
        if not v.get("CFG"):
            raise ValueError("the class-conditional scripts need CFG=<scale> in the argument file")
        if ddp:
            argv += ["--compute_fid", "--output_log", "{}_{}_{}{}_cfg{}.log".format(v["EXP"], v["EPOCH_ID"], v["METHOD"], v["STEPS"], v["CFG"])]
        else:
            argv += ["--master_port", "12004"]
        return argv
    v.setdefault("CH_MULT", "1 2 3 4")
    v.setdefault("ATTN_RES", "16 8 4")
    v.setdefault("Bs", "100")
    v.setdefault("BASE_CH", "256")
    argv = ["--exp", v["EXP"], "--dataset", v["DATASET"], "--batch_size", v["Bs"], "--epoch_id", v["EPOCH_ID"],
            "--image_size", v["IMG_SIZE"], "--f", "8", "--num_in_channels", "4", "--num_out_channels", "4",
            "--nf", v["BASE_CH"], "--ch_mult", *v["CH_MULT"].split(), "--attn_resolution", *v["ATTN_RES"].split(), "--num_res_blocks", "2",
            "--method", v["METHOD"], "--num_steps", v["STEPS"], "--model_type", v["MODEL_TYPE"]]
    if ddp:
        argv += ["--compute_fid", "--output_log", "{}_{}_{}{}.log".format(v["EXP"], v["EPOCH_ID"], v["METHOD"], v["STEPS"])]
    else:
        argv += ["--master_port", "12004", "--num_process_per_node", "1"]
    if v.get("USE_ORIGIN_ADM", "false") == "true":
        argv += ["--use_origin_adm"]
    else:
        argv += ["--num_classes", "1", "--label_dropout", "0."]
    return argv


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    extra = []
    if "--" in argv:
        k = argv.index("--")
        argv, extra = argv[:k], argv[k + 1:]
    p = argparse.ArgumentParser("lfm_amd.run_test")
    p.add_argument("args_file")
    p.add_argument("--cls", action="store_true", help="run_test_cls*.sh (ImageNet class-conditional, CFG)")
    p.add_argument("--ddp", action="store_true", help="run_test*_ddp.sh (one process per GPU through torch.distributed.run, --compute_fid)")
    p.add_argument("--num_gpus", type=int, default=8)
    p.add_argument("--dry_run", action="store_true", help="print the command instead of running it")
    a = p.parse_args(argv)
    print("Argument file: {}".format(a.args_file))
    text = open(a.args_file).read()
    print(text)
    flags = build_argv(parse_args_file(text), cls=a.cls, ddp=a.ddp) + extra
    if a.ddp:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.num_gpus), "--master-addr", "127.0.0.1",
               "-m", "lfm_amd.test_flow_latent_ddp", *flags]
        if a.dry_run:
            print(" ".join(shlex.quote(c) for c in cmd))
            return 0
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        return subprocess.call(cmd, env=env)
    if a.dry_run:
        print(" ".join(shlex.quote(c) for c in [sys.executable, "-m", "lfm_amd.test_flow_latent", *flags]))
        return 0
    from . import test_flow_latent

    test_flow_latent.main(flags)
    return 0


if __name__ == "__main__":
    sys.exit(main())

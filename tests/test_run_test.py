"""lfm_amd.run_test reproduces the flag expansion of the reference's launcher scripts for every argument file it ships
(bash_scripts/run_test{,_cls,_ddp,_cls_ddp}.sh, test_args/*.txt -- restated here verbatim: /root/reference is not on the GPU box)."""
import pytest

from lfm_amd import run_test
from lfm_amd.test_flow_latent import build_parser

ARGS = {
    "bed_adm": 'MODEL_TYPE=adm\nEPOCH_ID=425\nDATASET=lsun_bedroom\nEXP=bed_f8_adm\nMETHOD=dopri5\nSTEPS=0\nUSE_ORIGIN_ADM=false\nIMG_SIZE=256\nCH_MULT="1 2 3 4"\nATTN_RES="16 8 4"\n',
    "bed_dit": "MODEL_TYPE=DiT-L/2\nEPOCH_ID=550\nDATASET=lsun_bedroom\nEXP=bed_f8_dit\nMETHOD=dopri5\nSTEPS=0\nUSE_ORIGIN_ADM=false\nIMG_SIZE=256",
    "celeb256_adm": 'MODEL_TYPE=adm\nEPOCH_ID=450\nDATASET=celeba_256\nEXP=celeb256_f8_adm\nMETHOD=dopri5\nSTEPS=0\nUSE_ORIGIN_ADM=true\nIMG_SIZE=256\nCH_MULT="1 2 2 2"\nATTN_RES="16 8"',
    "celeb256_dit": "MODEL_TYPE=DiT-L/2\nEPOCH_ID=475\nDATASET=celeba_256\nEXP=celeb_f8_dit\nMETHOD=dopri5\nSTEPS=0\nUSE_ORIGIN_ADM=false\nIMG_SIZE=256",
    "celeb512_adm": 'MODEL_TYPE=adm\nEPOCH_ID=425\nDATASET=celeba_512\nEXP=celeb512_f8_adm\nMETHOD=dopri5\nSTEPS=0\nUSE_ORIGIN_ADM=true\nIMG_SIZE=512\nCH_MULT="1 2 2 2 4"\nATTN_RES="16 8"\nBs=16',
    "church_adm": 'MODEL_TYPE=adm\nEPOCH_ID=425\nDATASET=lsun_church\nEXP=church_f8_adm\nMETHOD=dopri5\nSTEPS=0\nUSE_ORIGIN_ADM=true\nIMG_SIZE=256\nCH_MULT="1 2 3 4"\nATTN_RES="16 8"',
    "church_dit": "MODEL_TYPE=DiT-L/2\nEPOCH_ID=575\nDATASET=lsun_church\nEXP=church_f8_dit\nMETHOD=dopri5\nSTEPS=0\nUSE_ORIGIN_ADM=false\nIMG_SIZE=256",
    "ffhq_adm": 'MODEL_TYPE=adm\nEPOCH_ID=400\nDATASET=ffhq_256\nEXP=ffhq_f8_adm\nMETHOD=dopri5\nSTEPS=0\nUSE_ORIGIN_ADM=false\nIMG_SIZE=256\nCH_MULT="1 2 3 4"\nATTN_RES="16 8 4"',
    "ffhq_dit": "MODEL_TYPE=DiT-L/2\nEPOCH_ID=475\nDATASET=ffhq_256\nEXP=ffhq_f8_dit\nMETHOD=dopri5\nSTEPS=0\nUSE_ORIGIN_ADM=false\nIMG_SIZE=256",
    "imnet_adm": 'MODEL_TYPE=adm\nEPOCH_ID=1125\nDATASET=imagenet_256\nEXP=imnet_f8_adm\nMETHOD=dopri5\nSTEPS=0\nCFG=1.25\nIMG_SIZE=256\nCH_MULT="1 2 3 4"\nATTN_RES="16 8 4"',
    "imnet_dit": "MODEL_TYPE=DiT-B/2\nEPOCH_ID=875\nDATASET=imagenet_256\nEXP=imnet_f8_ditb2\nMETHOD=dopri5\nSTEPS=0\nCFG=1.5",
}


def test_every_reference_args_file_parses_and_expands():
    for name, text in ARGS.items():
        cfg = run_test.parse_args_file(text)
        assert set(cfg) <= set(run_test.KEYS), name
        cls = "CFG" in cfg
        for ddp in (False, True):
            argv = run_test.build_argv(cfg, cls=cls, ddp=ddp)
            ns = build_parser().parse_args(argv)          # '--attn_resolution' is an argparse prefix of --attn_resolutions, as in the reference
            assert ns.exp == cfg["EXP"] and ns.dataset == cfg["DATASET"] and ns.epoch_id == int(cfg["EPOCH_ID"]) and ns.model_type == cfg["MODEL_TYPE"]
            assert ns.f == 8 and ns.num_in_channels == 4 and ns.num_out_channels == 4 and ns.num_res_blocks == 2
            assert ns.compute_fid == ddp
            if cls:
                assert (ns.num_classes, ns.label_dim, ns.label_dropout, ns.image_size, ns.batch_size) == (1000, 1000, 0.1, 256, 50)
                assert ns.cfg_scale == float(cfg["CFG"]) and list(ns.ch_mult) == [1, 2, 3, 4] and list(ns.attn_resolutions) == [16, 8, 4]
                if ddp:
                    assert ns.output_log == "{}_{}_dopri50_cfg{}.log".format(cfg["EXP"], cfg["EPOCH_ID"], cfg["CFG"])
            else:
                assert ns.image_size == int(cfg["IMG_SIZE"]) and ns.batch_size == int(cfg.get("Bs", 100)) and ns.nf == 256
                assert list(ns.ch_mult) == [int(x) for x in cfg.get("CH_MULT", "1 2 3 4").split()]
                assert list(ns.attn_resolutions) == [int(x) for x in cfg.get("ATTN_RES", "16 8 4").split()]
                if cfg["USE_ORIGIN_ADM"] == "true":
                    assert ns.use_origin_adm and ns.num_classes is None
                else:
                    assert not ns.use_origin_adm and ns.num_classes == 1 and ns.label_dropout == 0.0
                if ddp:
                    assert ns.output_log == "{}_{}_dopri50.log".format(cfg["EXP"], cfg["EPOCH_ID"])


def test_exact_flag_list_of_run_test_sh():
    cfg = run_test.parse_args_file(ARGS["celeb512_adm"])
    assert run_test.build_argv(cfg) == [
        "--exp", "celeb512_f8_adm", "--dataset", "celeba_512", "--batch_size", "16", "--epoch_id", "425", "--image_size", "512", "--f", "8",
        "--num_in_channels", "4", "--num_out_channels", "4", "--nf", "256", "--ch_mult", "1", "2", "2", "2", "4", "--attn_resolution", "16", "8",
        "--num_res_blocks", "2", "--method", "dopri5", "--num_steps", "0", "--model_type", "adm", "--master_port", "12004",
        "--num_process_per_node", "1", "--use_origin_adm"]


def test_args_file_errors_and_dry_run(tmp_path, capsys):
    with pytest.raises(ValueError):
        run_test.parse_args_file("MODEL_TYPE adm")
    with pytest.raises(ValueError):
        run_test.build_argv(run_test.parse_args_file(ARGS["celeb256_dit"]), cls=True)  # no CFG in an unconditional file
    assert run_test.parse_args_file('A="1 2"  # trailing comment\n\n# full-line comment\nB=3') == {"A": "1 2", "B": "3"}
    f = tmp_path / "imnet_dit.txt"
    f.write_text(ARGS["imnet_dit"])
    assert run_test.main([str(f), "--cls", "--ddp", "--num_gpus", "2", "--dry_run", "--", "--random_weights"]) == 0
    out = capsys.readouterr().out
    assert "torch.distributed.run" in out and "--nproc-per-node 2" in out and "lfm_amd.test_flow_latent_ddp" in out
    assert "--cfg_scale 1.5" in out and out.rstrip().endswith("--random_weights")

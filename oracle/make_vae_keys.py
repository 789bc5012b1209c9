"""Writes tests/golden/vae_decoder_keys.json (+ vae_encoder_keys.json): the state-dict keys and shapes of diffusers' ``AutoencoderKL`` for the
``stabilityai/sd-vae-ft-mse`` config, spelled out from the published module tree (diffusers >= 0.2x naming: ``to_q / to_k / to_v /
to_out.0``; config: latent_channels 4, block_out_channels [128, 256, 512, 512], layers_per_block 2, norm_num_groups 32).
Deliberately independent of oracle/vae_ref.py and lfm_amd/autoencoder.py (test infrastructure; diffusers itself is not installable)."""
import json
import os

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "vae_decoder_keys.json")


def main():
    keys = {}

    def conv(name, cout, cin, k):
        keys[name + ".weight"] = [cout, cin, k, k]
        keys[name + ".bias"] = [cout]

    def vec(name, c):
        keys[name + ".weight"] = [c]
        keys[name + ".bias"] = [c]

    def lin(name, cout, cin):
        keys[name + ".weight"] = [cout, cin]
        keys[name + ".bias"] = [cout]

    def resnet(name, cin, cout):
        vec(name + ".norm1", cin)
        conv(name + ".conv1", cout, cin, 3)
        vec(name + ".norm2", cout)
        conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(name + ".conv_shortcut", cout, cin, 1)

    conv("post_quant_conv", 4, 4, 1)
    conv("decoder.conv_in", 512, 4, 3)
    resnet("decoder.mid_block.resnets.0", 512, 512)
    vec("decoder.mid_block.attentions.0.group_norm", 512)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        lin("decoder.mid_block.attentions.0." + n, 512, 512)
    resnet("decoder.mid_block.resnets.1", 512, 512)
    # up_blocks run over reversed(block_out_channels) = 512, 512, 256, 128; three resnets each; an upsampler on all but the last
    resnet("decoder.up_blocks.0.resnets.0", 512, 512)
    resnet("decoder.up_blocks.0.resnets.1", 512, 512)
    resnet("decoder.up_blocks.0.resnets.2", 512, 512)
    conv("decoder.up_blocks.0.upsamplers.0.conv", 512, 512, 3)
    resnet("decoder.up_blocks.1.resnets.0", 512, 512)
    resnet("decoder.up_blocks.1.resnets.1", 512, 512)
    resnet("decoder.up_blocks.1.resnets.2", 512, 512)
    conv("decoder.up_blocks.1.upsamplers.0.conv", 512, 512, 3)
    resnet("decoder.up_blocks.2.resnets.0", 512, 256)
    resnet("decoder.up_blocks.2.resnets.1", 256, 256)
    resnet("decoder.up_blocks.2.resnets.2", 256, 256)
    conv("decoder.up_blocks.2.upsamplers.0.conv", 256, 256, 3)
    resnet("decoder.up_blocks.3.resnets.0", 256, 128)
    resnet("decoder.up_blocks.3.resnets.1", 128, 128)
    resnet("decoder.up_blocks.3.resnets.2", 128, 128)
    vec("decoder.conv_norm_out", 128)
    conv("decoder.conv_out", 3, 128, 3)
    json.dump(keys, open(OUT, "w"), indent=0, sort_keys=True)
    print(len(keys), "keys ->", OUT)

    # ---- encoder half + quant_conv: down_blocks over block_out_channels = 128, 256, 512, 512; TWO resnets each (layers_per_block);
    # a stride-2 downsampler conv on all but the last; mid block as in the decoder; double_z => conv_out has 2 * latent_channels outputs
    keys.clear()
    conv("quant_conv", 8, 8, 1)
    conv("encoder.conv_in", 128, 3, 3)
    resnet("encoder.down_blocks.0.resnets.0", 128, 128)
    resnet("encoder.down_blocks.0.resnets.1", 128, 128)
    conv("encoder.down_blocks.0.downsamplers.0.conv", 128, 128, 3)
    resnet("encoder.down_blocks.1.resnets.0", 128, 256)
    resnet("encoder.down_blocks.1.resnets.1", 256, 256)
    conv("encoder.down_blocks.1.downsamplers.0.conv", 256, 256, 3)
    resnet("encoder.down_blocks.2.resnets.0", 256, 512)
    resnet("encoder.down_blocks.2.resnets.1", 512, 512)
    conv("encoder.down_blocks.2.downsamplers.0.conv", 512, 512, 3)
    resnet("encoder.down_blocks.3.resnets.0", 512, 512)
    resnet("encoder.down_blocks.3.resnets.1", 512, 512)
    resnet("encoder.mid_block.resnets.0", 512, 512)
    vec("encoder.mid_block.attentions.0.group_norm", 512)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        lin("encoder.mid_block.attentions.0." + n, 512, 512)
    resnet("encoder.mid_block.resnets.1", 512, 512)
    vec("encoder.conv_norm_out", 512)
    conv("encoder.conv_out", 8, 512, 3)
    out2 = OUT.replace("vae_decoder_keys", "vae_encoder_keys")
    json.dump(keys, open(out2, "w"), indent=0, sort_keys=True)
    print(len(keys), "keys ->", out2)


if __name__ == "__main__":
    main()

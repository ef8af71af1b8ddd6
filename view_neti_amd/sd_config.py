"""Architecture descriptions of the frozen Stable-Diffusion parts the TI train step runs through
(UNet2DConditionModel, AutoencoderKL encoder, CLIP text encoder, DDPM scheduler) and the
state-dict key/shape enumeration shared by the synthetic-weight generator, the HIP engine and
the CPU oracle.  Key names are the diffusers 0.14 / transformers 4.27 ones so
that real checkpoints can be dropped in by name.

Reference: the modules are instantiated in training/coach.py:600-640 from
`cfg.model.pretrained_model_name_or_path` (training/config.py:82).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Tuple


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    # diffusers quirk: for SD `attention_head_dim` is the NUMBER of heads per block
    num_heads: Tuple[int, ...] = (8, 8, 8, 8)
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    use_linear_projection: bool = False
    # which down blocks carry transformers (SD: first three)
    down_has_attn: Tuple[bool, ...] = (True, True, True, False)

    @property
    def temb_dim(self) -> int:
        return self.block_out_channels[0] * 4

    @property
    def n_cross_layers(self) -> int:
        n = sum(self.layers_per_block for a in self.down_has_attn if a) + 1
        n += sum(self.layers_per_block + 1 for a in reversed(self.down_has_attn) if a)
        return n


@dataclass
class VAEConfig:
    in_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    scaling_factor: float = 0.18215


@dataclass
class CLIPTextConfig:
    vocab_size: int = 49408
    hidden_size: int = 768
    num_layers: int = 12
    num_heads: int = 12
    intermediate_size: int = 3072
    max_positions: int = 77
    act: str = "quick_gelu"  # "gelu" for the OpenCLIP-H encoder of SD-2.x
    eps: float = 1e-5


@dataclass
class DDPMConfig:
    num_train_timesteps: int = 1000
    beta_start: float = 0.00085
    beta_end: float = 0.012
    prediction_type: str = "epsilon"


@dataclass
class SDConfig:
    name: str = "sd15"
    unet: UNetConfig = field(default_factory=UNetConfig)
    vae: VAEConfig = field(default_factory=VAEConfig)
    clip: CLIPTextConfig = field(default_factory=CLIPTextConfig)
    ddpm: DDPMConfig = field(default_factory=DDPMConfig)


def sd15() -> SDConfig:
    """SD-1.4/1.5 shape family (BASELINE.json headline config)."""
    return SDConfig()


def sd21() -> SDConfig:
    """SD-2.1 shape family used by every shipped YAML of the reference (input_configs/*.yaml)."""
    return SDConfig(
        name="sd21",
        unet=UNetConfig(num_heads=(5, 10, 20, 20), cross_attention_dim=1024, use_linear_projection=True),
        clip=CLIPTextConfig(hidden_size=1024, num_layers=23, num_heads=16, intermediate_size=4096, act="gelu"),
        ddpm=DDPMConfig(prediction_type="v_prediction"),
    )


def tiny() -> SDConfig:
    """A structurally identical but small configuration for parity tests the CPU oracle can run
    in seconds: same block graph, 16 cross-attention layers, head_dim 64."""
    return SDConfig(
        name="tiny",
        unet=UNetConfig(block_out_channels=(64, 128, 256, 256), num_heads=(1, 2, 4, 4), cross_attention_dim=128,
                        norm_num_groups=16),
        vae=VAEConfig(block_out_channels=(64, 128, 128, 128), norm_num_groups=16),
        clip=CLIPTextConfig(vocab_size=1024, hidden_size=128, num_layers=2, num_heads=2, intermediate_size=256),
    )


def tiny21() -> SDConfig:
    """`tiny` with the SD-2.1 family's structural switches: linear proj_in/out, exact-GELU CLIP,
    v-prediction target (BASELINE.json configs 3-5 run on SD-2.1 shapes)."""
    c = tiny()
    return SDConfig(
        name="tiny21",
        unet=UNetConfig(block_out_channels=c.unet.block_out_channels, num_heads=c.unet.num_heads,
                        cross_attention_dim=c.unet.cross_attention_dim, norm_num_groups=c.unet.norm_num_groups,
                        use_linear_projection=True),
        vae=c.vae,
        clip=CLIPTextConfig(vocab_size=1024, hidden_size=128, num_layers=3, num_heads=2, intermediate_size=256,
                            act="gelu"),
        ddpm=DDPMConfig(prediction_type="v_prediction"),
    )


CONFIGS = {"sd15": sd15, "sd21": sd21, "tiny": tiny, "tiny21": tiny21}


# ----------------------------------------------------------------------------------------------
# state-dict enumeration
# ----------------------------------------------------------------------------------------------
Shapes = Dict[str, Tuple[int, ...]]


def _resnet(s: Shapes, p: str, cin: int, cout: int, temb: int | None):
    s[p + "norm1.weight"] = (cin,)
    s[p + "norm1.bias"] = (cin,)
    s[p + "conv1.weight"] = (cout, cin, 3, 3)
    s[p + "conv1.bias"] = (cout,)
    if temb is not None:
        s[p + "time_emb_proj.weight"] = (cout, temb)
        s[p + "time_emb_proj.bias"] = (cout,)
    s[p + "norm2.weight"] = (cout,)
    s[p + "norm2.bias"] = (cout,)
    s[p + "conv2.weight"] = (cout, cout, 3, 3)
    s[p + "conv2.bias"] = (cout,)
    if cin != cout:
        s[p + "conv_shortcut.weight"] = (cout, cin, 1, 1)
        s[p + "conv_shortcut.bias"] = (cout,)


def _transformer(s: Shapes, p: str, c: int, ctx: int, linear_proj: bool):
    s[p + "norm.weight"] = (c,)
    s[p + "norm.bias"] = (c,)
    proj_shape = (c, c) if linear_proj else (c, c, 1, 1)
    s[p + "proj_in.weight"] = proj_shape
    s[p + "proj_in.bias"] = (c,)
    t = p + "transformer_blocks.0."
    for n in ("norm1", "norm2", "norm3"):
        s[t + n + ".weight"] = (c,)
        s[t + n + ".bias"] = (c,)
    for a, kd in (("attn1", c), ("attn2", ctx)):
        s[t + a + ".to_q.weight"] = (c, c)
        s[t + a + ".to_k.weight"] = (c, kd)
        s[t + a + ".to_v.weight"] = (c, kd)
        s[t + a + ".to_out.0.weight"] = (c, c)
        s[t + a + ".to_out.0.bias"] = (c,)
    s[t + "ff.net.0.proj.weight"] = (8 * c, c)
    s[t + "ff.net.0.proj.bias"] = (8 * c,)
    s[t + "ff.net.2.weight"] = (c, 4 * c)
    s[t + "ff.net.2.bias"] = (c,)
    s[p + "proj_out.weight"] = proj_shape
    s[p + "proj_out.bias"] = (c,)


def up_block_channels(cfg: UNetConfig, i: int, j: int) -> Tuple[int, int, int]:
    """(resnet_in, skip, out) channels of resnet j of up block i (diffusers get_up_block logic)."""
    rev = tuple(reversed(cfg.block_out_channels))
    out = rev[i]
    prev = rev[i - 1] if i > 0 else rev[0]
    inp = rev[min(i + 1, len(rev) - 1)]
    n = cfg.layers_per_block + 1
    skip = inp if j == n - 1 else out
    rin = prev if j == 0 else out
    return rin, skip, out


def unet_shapes(cfg: UNetConfig) -> Shapes:
    s: Shapes = {}
    boc = cfg.block_out_channels
    temb = cfg.temb_dim
    s["conv_in.weight"] = (boc[0], cfg.in_channels, 3, 3)
    s["conv_in.bias"] = (boc[0],)
    s["time_embedding.linear_1.weight"] = (temb, boc[0])
    s["time_embedding.linear_1.bias"] = (temb,)
    s["time_embedding.linear_2.weight"] = (temb, temb)
    s["time_embedding.linear_2.bias"] = (temb,)
    cin = boc[0]
    for i, cout in enumerate(boc):
        for j in range(cfg.layers_per_block):
            _resnet(s, f"down_blocks.{i}.resnets.{j}.", cin if j == 0 else cout, cout, temb)
            if cfg.down_has_attn[i]:
                _transformer(s, f"down_blocks.{i}.attentions.{j}.", cout, cfg.cross_attention_dim,
                             cfg.use_linear_projection)
        if i < len(boc) - 1:
            s[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            s[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (cout,)
        cin = cout
    cm = boc[-1]
    _resnet(s, "mid_block.resnets.0.", cm, cm, temb)
    _transformer(s, "mid_block.attentions.0.", cm, cfg.cross_attention_dim, cfg.use_linear_projection)
    _resnet(s, "mid_block.resnets.1.", cm, cm, temb)
    up_has_attn = tuple(reversed(cfg.down_has_attn))
    for i in range(len(boc)):
        for j in range(cfg.layers_per_block + 1):
            rin, skip, out = up_block_channels(cfg, i, j)
            _resnet(s, f"up_blocks.{i}.resnets.{j}.", rin + skip, out, temb)
            if up_has_attn[i]:
                _transformer(s, f"up_blocks.{i}.attentions.{j}.", out, cfg.cross_attention_dim,
                             cfg.use_linear_projection)
        if i < len(boc) - 1:
            out = tuple(reversed(boc))[i]
            s[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (out, out, 3, 3)
            s[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (out,)
    s["conv_norm_out.weight"] = (boc[0],)
    s["conv_norm_out.bias"] = (boc[0],)
    s["conv_out.weight"] = (cfg.out_channels, boc[0], 3, 3)
    s["conv_out.bias"] = (cfg.out_channels,)
    return s


def vae_encoder_shapes(cfg: VAEConfig) -> Shapes:
    s: Shapes = {}
    boc = cfg.block_out_channels
    s["encoder.conv_in.weight"] = (boc[0], cfg.in_channels, 3, 3)
    s["encoder.conv_in.bias"] = (boc[0],)
    cin = boc[0]
    for i, cout in enumerate(boc):
        for j in range(cfg.layers_per_block):
            _resnet(s, f"encoder.down_blocks.{i}.resnets.{j}.", cin if j == 0 else cout, cout, None)
        if i < len(boc) - 1:
            s[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            s[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (cout,)
        cin = cout
    cm = boc[-1]
    _resnet(s, "encoder.mid_block.resnets.0.", cm, cm, None)
    a = "encoder.mid_block.attentions.0."
    s[a + "group_norm.weight"] = (cm,)
    s[a + "group_norm.bias"] = (cm,)
    for n in ("query", "key", "value", "proj_attn"):
        s[a + n + ".weight"] = (cm, cm)
        s[a + n + ".bias"] = (cm,)
    _resnet(s, "encoder.mid_block.resnets.1.", cm, cm, None)
    s["encoder.conv_norm_out.weight"] = (cm,)
    s["encoder.conv_norm_out.bias"] = (cm,)
    s["encoder.conv_out.weight"] = (2 * cfg.latent_channels, cm, 3, 3)
    s["encoder.conv_out.bias"] = (2 * cfg.latent_channels,)
    s["quant_conv.weight"] = (2 * cfg.latent_channels, 2 * cfg.latent_channels, 1, 1)
    s["quant_conv.bias"] = (2 * cfg.latent_channels,)
    return s


def vae_decoder_shapes(cfg: VAEConfig) -> Shapes:
    """AutoencoderKL `post_quant_conv` + `decoder.*` (diffusers 0.14 key names)."""
    s: Shapes = {}
    boc = list(reversed(cfg.block_out_channels))
    lc = cfg.latent_channels
    s["post_quant_conv.weight"] = (lc, lc, 1, 1)
    s["post_quant_conv.bias"] = (lc,)
    cm = boc[0]
    s["decoder.conv_in.weight"] = (cm, lc, 3, 3)
    s["decoder.conv_in.bias"] = (cm,)
    _resnet(s, "decoder.mid_block.resnets.0.", cm, cm, None)
    a = "decoder.mid_block.attentions.0."
    s[a + "group_norm.weight"] = (cm,)
    s[a + "group_norm.bias"] = (cm,)
    for n in ("query", "key", "value", "proj_attn"):
        s[a + n + ".weight"] = (cm, cm)
        s[a + n + ".bias"] = (cm,)
    _resnet(s, "decoder.mid_block.resnets.1.", cm, cm, None)
    cin = cm
    for i, cout in enumerate(boc):
        for j in range(cfg.layers_per_block + 1):
            _resnet(s, f"decoder.up_blocks.{i}.resnets.{j}.", cin if j == 0 else cout, cout, None)
        if i < len(boc) - 1:
            s[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            s[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (cout,)
        cin = cout
    s["decoder.conv_norm_out.weight"] = (boc[-1],)
    s["decoder.conv_norm_out.bias"] = (boc[-1],)
    s["decoder.conv_out.weight"] = (cfg.in_channels, boc[-1], 3, 3)
    s["decoder.conv_out.bias"] = (cfg.in_channels,)
    return s


def clip_text_shapes(cfg: CLIPTextConfig) -> Shapes:
    s: Shapes = {}
    d, f = cfg.hidden_size, cfg.intermediate_size
    s["text_model.embeddings.token_embedding.weight"] = (cfg.vocab_size, d)
    s["text_model.embeddings.position_embedding.weight"] = (cfg.max_positions, d)
    for i in range(cfg.num_layers):
        p = f"text_model.encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"] = (d, d)
            s[p + f"self_attn.{n}.bias"] = (d,)
        for n in ("layer_norm1", "layer_norm2"):
            s[p + n + ".weight"] = (d,)
            s[p + n + ".bias"] = (d,)
        s[p + "mlp.fc1.weight"] = (f, d)
        s[p + "mlp.fc1.bias"] = (f,)
        s[p + "mlp.fc2.weight"] = (d, f)
        s[p + "mlp.fc2.bias"] = (d,)
    s["text_model.final_layer_norm.weight"] = (d,)
    s["text_model.final_layer_norm.bias"] = (d,)
    return s


def cross_attention_order(cfg: UNetConfig) -> List[str]:
    """Module prefixes of the transformers in UNet call order == reference constants.UNET_LAYERS
    order (constants.py:1-4): down (i,j)..., mid, up (i,j)..."""
    order = []
    for i, has in enumerate(cfg.down_has_attn):
        if has:
            order += [f"down_blocks.{i}.attentions.{j}." for j in range(cfg.layers_per_block)]
    order.append("mid_block.attentions.0.")
    for i, has in enumerate(reversed(cfg.down_has_attn)):
        if has:
            order += [f"up_blocks.{i}.attentions.{j}." for j in range(cfg.layers_per_block + 1)]
    return order

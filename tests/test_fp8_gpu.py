"""BASELINE.json configs[4] — fp8 (OCP e4m3) operands on CDNA4's fp8 matrix instruction — for the UNet's 3x3 convolutions
(gill_amd/csrc/conv_fp8.hip).  No reference counterpart exists (the reference runs SD in fp16): the kernel is checked against an
fp32 convolution of the SAME quantised operands (torch.float8_e4m3fn emulates the quantiser on the CPU; bar: fp32-accumulation
noise), and the whole fp8 UNet mode against this build's own bf16 path (accuracy delta stated in the test)."""
import pytest
import torch
import torch.nn.functional as F

from gill_amd import synth

pytestmark = pytest.mark.gpu
ACT_SCALE = 8.0


def _q(t, scale):
  """fp8 e4m3 round trip of t * scale (round-to-nearest-even, saturating at 448), back in fp32 and un-scaled."""
  return (t * scale).clamp(-448, 448).to(torch.float8_e4m3fn).float() / scale


def _ref(x_nhwc, w, bias, resid):
  xq = _q(x_nhwc.float(), ACT_SCALE)
  s = w.abs().amax(dim=(1, 2, 3), keepdim=True) / 448.0
  wq = _q(w / s, 1.0) * s
  y = F.conv2d(xq.permute(0, 3, 1, 2), wq, bias, padding=1).permute(0, 2, 3, 1)
  if resid is not None:
    y = y + resid.float()
  return y


@pytest.mark.parametrize("B,H,W,Cin,Cout,sk", [
  (2, 16, 16, 128, 160, 1),      # even number of 64-channel halves per tap
  (2, 16, 16, 320, 320, 1),      # 9 * 5 = 45 halves: the last K step is half padding; two 160-wide N tiles
  (1, 8, 8, 64, 128, 1),         # one half per tap: every K step mixes two taps; BN = 128
  (3, 12, 20, 192, 64, 1),       # W not a power of two, ragged M (720 rows), N < tile
  (2, 16, 16, 640, 320, 4),      # split-K partials finished by the bf16 path's reducer
  (1, 8, 8, 1280, 1280, 16),
])
def test_conv3x3_fp8_vs_quantised_fp32(cuda, B, H, W, Cin, Cout, sk):
  from gill_amd import ops
  x = synth.normal("f8_x", (B, H, W, Cin), 1).bfloat16()
  x = torch.nn.functional.silu(x.float()).bfloat16()             # the operand's real distribution: SiLU of a unit normal
  w = synth.normal("f8_w", (Cout, Cin, 3, 3), 2, std=0.03)
  bias = synth.normal("f8_b", (Cout,), 3, std=0.1)
  resid = synth.normal("f8_r", (B, H, W, Cout), 4).bfloat16()
  got = ops.conv3x3_fp8(x.to(cuda), w.to(cuda), bias.to(cuda), resid.to(cuda), splitk=sk).float().cpu()
  ref = _ref(x, w, bias, resid)
  err = (got - ref).abs().max().item()
  scale = ref.abs().max().item()
  print(f"[conv fp8 {B}x{H}x{W}x{Cin}->{Cout} sk{sk}] max_abs={err:.3e} of {scale:.2f}")
  assert err < 1.5e-2 * scale      # bf16 output rounding (2^-9 relative) + fp32 accumulation order; the quantisation itself is in `ref`
  # and the quantisation error against the un-quantised convolution is what e4m3 operands cost (reported, loosely bounded)
  full = F.conv2d(x.float().permute(0, 3, 1, 2), w, bias, padding=1).permute(0, 2, 3, 1) + resid.float()
  rel = ((got - full).norm() / full.norm()).item()
  print(f"  vs un-quantised conv: rel-L2 {rel:.3e}")
  assert rel < 6e-2


@pytest.mark.parametrize("M,C", [(256, 640), (2048, 1280), (200, 128), (8192, 640)])
def test_geglu_fp8_vs_quantised_fp32(cuda, M, C):
  """linear_fp8.hip: the GEGLU projection (diffusers ff.net.0 on norm3's output) with e4m3 activations and weights on
  v_mfma_scale_f32_16x16x128_f8f6f4, against fp32 arithmetic on the SAME quantised operands — fp8(16 * LNhat(t)) with the statistics of the bf16
  rows, rows of bf16(gamma * W) quantised per output row, bias + beta . W^T in fp32 (torch.float8_e4m3fn emulates the quantiser) — and, reported,
  against the un-quantised GEGLU (what the e4m3 operands cost).  Shapes: UNet level 1 (C = 640, 5 K steps of 128), level 2 (1280), a ragged M, and the
  level-1 row count of the 8-sample batch."""
  from gill_amd import ops
  inner = 4 * C
  t = (synth.normal("g8_t", (M, C), 11) * 1.7 + 0.3).bfloat16()
  g = 1.0 + 0.2 * synth.normal("g8_g", (C,), 12)
  be = 0.1 * synth.normal("g8_be", (C,), 13)
  w = synth.normal("g8_w", (2 * inner, C), 14, std=C ** -0.5).bfloat16()
  b = 0.1 * synth.normal("g8_b", (2 * inner,), 15)
  got = ops.geglu_fp8(t.to(cuda), g.to(cuda), be.to(cuda), w.to(cuda), b.to(cuda)).float().cpu()
  assert torch.isfinite(got).all()
  tf = t.float()
  mean = tf.mean(-1, keepdim=True)
  var = (tf * tf).mean(-1, keepdim=True) - mean * mean          # (the kernel's E[x^2] - E[x]^2 form)
  tn = (tf - mean) * torch.rsqrt(var.clamp_min(0) + 1e-5)
  xq = _q(tn, 16.0)
  wf = (w.float() * g).bfloat16().float()                        # ln_fold_rows: gamma folded into the rows, rounded to bf16
  sc = wf.abs().amax(dim=1, keepdim=True) / 448.0
  wq = _q(wf / sc, 1.0) * sc
  bias = b + w.float() @ be                                      # beta . W^T on the un-folded rows
  proj = xq @ wq.T + bias
  h, gate = proj.chunk(2, dim=-1)
  ref = h * F.gelu(gate)
  err = (got - ref).abs().max().item()
  scale = ref.abs().max().item()
  print(f"[geglu fp8 {M}x{C}] max_abs={err:.3e} of {scale:.2f}")
  # bf16 output rounding + fp32 accumulation order + e4m3 roundings flipped by the last bit of rstd (one flip = 6 % of that element): measured
  # 0.9-1.4e-2 of the output range (profiles/r06_fp8_tests.log); the quantisation itself is in `ref`
  assert err < 2e-2 * scale
  proj_full = F.layer_norm(tf, (C,), g, be, 1e-5) @ w.float().T + b
  hf, gf = proj_full.chunk(2, dim=-1)
  full = hf * F.gelu(gf)
  rel = ((got - full).norm() / full.norm()).item()
  print(f"  vs un-quantised GEGLU: rel-L2 {rel:.3e}")
  assert rel < 6e-2


def _bfw(sd):
  return {k: v.bfloat16().float() for k, v in sd.items()}


@pytest.mark.parametrize("which", ["tiny", "sd15"])
def test_fp8_unet_mode_vs_bf16_mode(cuda, which):
  """gill_unet_config.fp8_convs = 1 against this build's parity (bf16) configuration on the same weights and inputs through
  gill_sd_denoise (10 PLMS steps, CFG 7.5): the latents must stay within a stated distance — what the e4m3 operands of the 44
  resnet convolutions and (round 6) the 11 GEGLU projections of levels 1-3 cost after 11 recurrent UNet calls — and the fp8 mode must itself be bit-reproducible (its statistics
  and split-K reductions are the same fixed-order ones)."""
  import dataclasses
  from gill_amd.sd import GillSDPipeline
  cfg = synth.UNetConfig.tiny(16) if which == "tiny" else synth.UNetConfig.sd15()
  sd = _bfw(synth.unet_state_dict(cfg, seed=41))
  uncond = synth.uncond_context(cfg.ctx_len, cfg.cross_attention_dim, seed=41).bfloat16().float()
  B = 2
  cond = synth.normal("f8_cond", (B, cfg.ctx_len, cfg.cross_attention_dim), 42).bfloat16().float()
  lat0 = synth.initial_latents(B, 4, cfg.sample_size, seed=4242)
  kw = dict(prompt_embeds=cond, latents=lat0, guidance_scale=7.5, num_inference_steps=10)
  ref = GillSDPipeline(sd, cfg, uncond, cuda, max_batch=2 * B)(**kw).images.float().cpu()
  pipe8 = GillSDPipeline(sd, dataclasses.replace(cfg, fp8_convs=True), uncond, cuda, max_batch=2 * B)
  got = pipe8(**kw).images.float().cpu()
  assert torch.isfinite(got).all()
  rel = ((got - ref).norm() / ref.norm()).item()
  cos = torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0).item()
  print(f"[fp8 UNet mode vs bf16 mode, {which}, 10-step CFG] rel-L2 {rel:.3e} cos {cos:.5f}")
  # sd15: the bar the bf16 mode is held to against the fp32 oracle.  tiny: 1e-1 since round 6 — with the GEGLU projections in e4m3 as well (an e4m3
  # x e4m3 dot product of zero-mean operands carries ~5 % relative error whatever its length: test_geglu_fp8_vs_quantised_fp32 reports 5.3e-2) the
  # random tiny UNet lands at 8.3e-2 after its 11 recurrent calls; the full-size UNet at 6.0e-2 (profiles/r06_fp8_tests.log)
  assert rel < (1e-1 if which == "tiny" else 8e-2) and cos > 0.995
  again = pipe8(**kw).images.float().cpu()
  assert torch.equal(again, got)


@pytest.mark.parametrize("which", ["tiny", "sd15"])
def test_fp8_unet_forward_vs_fp32_oracle(cuda, which):
  """the fp8 mode was only ever compared with this build's own bf16 mode.  Here ONE UNet forward (CFG pair, t = 961) with
  the 44 resnet convolutions in e4m3 goes against the fp32 CPU oracle (oracle/unet_ref.py, the same one the bf16 mode is held to), next to
  the bf16 mode's own distance on the same inputs.  Measured: bf16 1.16e-2 / 1.21e-2; fp8 5.84e-2 at full size, 6.13e-2 on the tiny UNet, cosine
  0.9981-0.9983 (round 5, convolutions only: 5.4e-2 / 5.8e-2) — what e4m3 operands in the 44 resnet convolutions and the 11 GEGLU projections cost one forward.  Bars: bf16 5e-2 / 0.998 (its bar everywhere), fp8 8e-2 /
  0.997 (the bar bench.py's forward_check applies to the C5 line)."""
  import dataclasses
  from gill_amd.sd import GillSDPipeline
  from oracle import unet_ref
  cfg = synth.UNetConfig.tiny(16) if which == "tiny" else synth.UNetConfig.sd15()
  sd = _bfw(synth.unet_state_dict(cfg, seed=43))
  uncond = synth.uncond_context(cfg.ctx_len, cfg.cross_attention_dim, seed=43).bfloat16().float()
  L = cfg.sample_size
  x = synth.initial_latents(2, 4, L, seed=4343)
  ctx = torch.cat([uncond, synth.normal("f8o_ctx", (1, cfg.ctx_len, cfg.cross_attention_dim), 44)], 0).bfloat16().float()
  t = torch.tensor([961.0, 961.0])
  heads = cfg.heads_per_level if cfg.heads_per_level else cfg.num_heads
  ref = unet_ref.unet_forward(sd, x, t, ctx, cfg.block_out_channels, heads, cfg.norm_num_groups)
  out = {}
  for name, c in (("bf16", cfg), ("fp8", dataclasses.replace(cfg, fp8_convs=True))):
    got = GillSDPipeline(sd, c, uncond, cuda, max_batch=2).unet(x, t, ctx).float().cpu()
    rel = ((got - ref).norm() / ref.norm()).item()
    cos = torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0).item()
    print(f"[{name} UNet forward vs fp32 oracle, {which}] rel-L2 {rel:.3e} cos {cos:.5f}")
    out[name] = (rel, cos)
  assert out["bf16"][0] < 5e-2 and out["bf16"][1] > 0.998
  assert out["fp8"][0] < 8e-2 and out["fp8"][1] > 0.997

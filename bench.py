#!/usr/bin/env python
"""Benchmark of the GILL image-generation hot path on MI355X (contract: see the task brief / DESIGN.md section 6).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: one rank per GPU.  Under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` the ranks read
  RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; a plain `python bench.py --gpus N` re-executes itself under torch.distributed.run
  on 127.0.0.1 with a free port.)

A "step" = one pass of the hot path over one batch of synthetic prompts: token ids -> frozen OPT-6.7B forward ->
8 [IMG] hidden states -> GILLMapper -> (B,77,768) -> SD-1.5 UNet CFG/PLMS loop (50 steps = 51 UNet calls of batch 2B)
-> final latents, all-gathered over ranks -> VAE decode of the local shard to uint8 512x512.  Workload: N = 1 is
BASELINE.json configs[1] (4 prompts on the GPU); N > 1 is configs[2] (8 prompts per GPU: batch 64 over 8), weak scaling
over N >= 2; --prompts-per-gpu overrides.  Inputs (ids, weights, initial latents) are resident in HBM before the timed
region.  EVERY timed step's latents are kept and checked after the timed region (finite, and within REL_STEP_TOL of the
first step's: the workload repeats the same inputs); a violation prints the per-step table and exits non-zero.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time


import torch                      # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import gill_amd                   # noqa: E402
gill_amd.configure_hip_runtime()  # entry point: opt in to the process-wide hipGraph setting before HIP initialises (see its docstring)

UNET_TFLOP_PER_SAMPLE_FORWARD = 0.8032   # 401.6 GMAC, SD-1.5, 64x64 latents
UNET_TFLOP_SD21_768 = 2.149              # 1074.6 GMAC, SD-2.1-768, 96x96 latents (--config c4)
PEAK_BF16_TFLOPS = 2500.0                # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
REL_STEP_TOL = 1e-6                      # rel-L2 of a step's final latents vs the first timed step's: same inputs every step and
                                         # fixed-order reductions everywhere, so the steps are bit-identical (0 expected)


def gpu_state_dict(builder, cfg, dev, seed):
  """Shapes/names from gill_amd.synth, values drawn on the GPU (fan-in scaled like synth; perf is value-independent
  but random non-zero operands keep the DVFS behaviour honest)."""
  import numpy as np
  proto = builder(cfg, meta=True)
  g = torch.Generator(device=dev).manual_seed(seed)
  sd = {}
  for k, shape in proto.items():
    if k.endswith(".weight") and len(shape) >= 2:
      std = 1.0 / float(np.sqrt(np.prod(shape[1:])))
      if "embed_tokens" in k:
        std = 0.5
      elif "embed_positions" in k:
        std = 0.25
      t = torch.randn(shape, device=dev, dtype=torch.float32, generator=g).mul_(std).to(torch.bfloat16)
    elif k.endswith(".weight"):   # norm scales
      t = (1.0 + 0.1 * torch.randn(shape, device=dev, generator=g)).to(torch.bfloat16)
    elif k == "query_embs":
      t = torch.randn(shape, device=dev, generator=g).to(torch.bfloat16)
    else:
      t = (0.02 * torch.randn(shape, device=dev, generator=g)).to(torch.bfloat16)
    sd[k] = t
  return sd


def shapes_of(builder_name, cfg):
  """state-dict inventory (name -> shape) without materialising values: run the synth builder with a stub generator."""
  from gill_amd import synth
  real = synth.normal
  shapes = {}

  class _Fake:
    def __init__(self, shape):
      self.shape = tuple(shape)

  def fake_normal(name, shape, seed, std=1.0, mean=0.0):
    shapes[name] = tuple(shape)
    return _Fake(shape)
  synth.normal = fake_normal
  try:
    getattr(synth, builder_name)(cfg, 0)
  finally:
    synth.normal = real
  return shapes


def build_model(dev, opt_cfg, unet_cfg, max_prompts, vae_cfg=None, keep_unet_cpu=False):
  from types import SimpleNamespace
  from gill_amd import synth
  from gill_amd.models import GILL
  from gill_amd.sd import GillSDPipeline
  tok = synth.HashTokenizer()
  opt_sd = gpu_state_dict(lambda c, meta: shapes_of("opt_state_dict", c), opt_cfg, dev, 0)
  unet_sd = gpu_state_dict(lambda c, meta: shapes_of("unet_state_dict", c), unet_cfg, dev, 1)
  uncond = synth.uncond_context(unet_cfg.ctx_len, unet_cfg.cross_attention_dim, 0)
  vae_cfg = vae_cfg or synth.VAEConfig.sd15()
  vae_sd = gpu_state_dict(lambda c, meta: shapes_of("vae_decoder_state_dict", c), vae_cfg, dev, 3)
  pipe = GillSDPipeline(unet_sd, unet_cfg, uncond, dev, max_batch=2 * min(8, max_prompts), vae_state=vae_sd, vae_cfg=vae_cfg)
  # fp32 host copy of the very weights the handle holds (bf16 values, exact): what the cpu_baseline leg's oracle forward runs on
  pipe.bench_unet_sd_cpu = {k: v.float().cpu() for k, v in unet_sd.items()} if keep_unet_cpu else None
  del unet_sd, vae_sd
  name = "facebook/opt-6.7b" if opt_cfg.hidden_size == 4096 else "facebook/opt-125m"
  args = SimpleNamespace(freeze_lm=True, freeze_vm=True, opt_version=name, visual_encoder="openai/clip-vit-large-patch14",
                         n_visual_tokens=4, ret_emb_dim=256, gen_emb_dim=unet_cfg.cross_attention_dim, text_emb_layers=[-1], text_fc_mode="gill_mapper",
                         ret_text_fc_mode="linear", num_tokens=8, num_clip_tokens=77, retrieval_token_idx=synth.IMG_TOKEN_IDS,
                         gen_token_idx=synth.IMG_TOKEN_IDS, opt_state_dict=opt_sd)
  g = GILL(tok, args, load_sd=True, sd_pipe=pipe)
  msd = gpu_state_dict(lambda c, meta: shapes_of("mapper_state_dict", c),
                       synth.MapperConfig(in_dim=opt_cfg.hidden_size, out_dim=unet_cfg.cross_attention_dim), dev, 2)
  g.model.gen_text_hidden_fcs[0].load_state_dict({k: v.float().cpu() for k, v in msd.items()}, strict=True)
  g = g.eval().bfloat16().cuda(dev)
  return g


SETUP_KERNELS = ("at::native", "__amd_rocclr", "convert_", "relayout", "pad_head", "scatter_rows", "permute", "ln_fold",
                 "vec_add", "cast_", "ffo_fuse", "kperm")   # weight set-up (create time) / torch's own kernels: not the hot path


def pmc_traffic_live(a):
  """HBM bytes of ONE step of this very workload, measured now: two rocprofv3 passes (--pmc FETCH_SIZE, then --pmc WRITE_SIZE;
  counters only, no trace domains) over a child `bench.py --steps 1 --warmup 0 --child-pmc` with the same prompts / steps, summed
  over every dispatch of the hot-path kernels and corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE on gfx950 tallies
  128-B requests as 64 B: x2; both counters are in KiB).  The child runs the forward eagerly (GILL_NO_GRAPH=1) so that every
  kernel is a dispatch of its own for the counter service.  Returns (GB per step, detail dict) or (None, reason)."""
  import csv
  import glob
  import shutil
  import tempfile
  rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
  if not os.path.exists(rocprof):
    return None, "rocprofv3 not found"
  tot = {}
  t0 = time.time()
  for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    d = tempfile.mkdtemp(prefix=f"gill_pmc_{ctr}_", dir="/tmp")
    env = dict(os.environ, GILL_NO_GRAPH="1", TMPDIR="/tmp")
    cmd = [rocprof, "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
           "--steps", "1", "--warmup", "0", "--child-pmc", "--prompts-per-gpu", str(a.prompts_per_gpu),
           "--infer-steps", str(a.infer_steps), "--prompt-len", str(a.prompt_len)]
    try:
      r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=a.pmc_timeout)
    except subprocess.TimeoutExpired:
      shutil.rmtree(d, ignore_errors=True)
      return None, f"rocprofv3 --pmc {ctr} pass exceeded {a.pmc_timeout} s"
    kb, n = 0.0, 0
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
      with open(f) as fh:
        for row in csv.DictReader(fh):
          if row["Counter_Name"] != ctr or any(t in row["Kernel_Name"] for t in SETUP_KERNELS):
            continue
          kb += float(row["Counter_Value"])
          n += 1
    shutil.rmtree(d, ignore_errors=True)
    if r.returncode != 0 or n == 0:
      return None, f"rocprofv3 --pmc {ctr} pass failed (rc {r.returncode}, {n} dispatches): {r.stdout.decode(errors='replace')[-300:]}"
    tot[ctr] = (kb, n)
  gb = (tot["FETCH_SIZE"][0] * 2.0 + tot["WRITE_SIZE"][0]) * 1024.0 / 1e9
  return gb, {"FETCH_SIZE_KiB_raw": tot["FETCH_SIZE"][0], "WRITE_SIZE_KiB": tot["WRITE_SIZE"][0], "gfx950_fetch_correction": 2.0,
              "dispatches": tot["FETCH_SIZE"][1], "passes_s": round(time.time() - t0, 1)}


def kernel_rooflines(dev):
  """The three dominant device kernels of the UNet loop, each timed alone with HIP events on its level-0 shape of the CFG
  batch 8 through the operator C ABI (gill_op_*): algorithmic FLOPs / average launch time."""
  from gill_amd import ops
  out = []
  R1, R2 = 5, 45      # GILL_OP_REPEAT is read by libgill_amd at every gill_op_* call: a call launches its kernel that many times

  def timed(fn, flops, name):
    # per-launch time = (call at R2 repeats - call at R1 repeats) / (R2 - R1): the wrapper's own work (weight re-layout, scratch
    # allocation, layout conversions) is in both calls and cancels
    ms = {}
    for r in (R1, R2, R1, R2):
      os.environ["GILL_OP_REPEAT"] = str(r)
      fn()
      torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record(); fn(); e1.record(); torch.cuda.synchronize()
      ms[r] = e0.elapsed_time(e1)
    os.environ["GILL_OP_REPEAT"] = "1"
    us = (ms[R2] - ms[R1]) * 1e3 / (R2 - R1)
    out.append({"kernel": name, "avg_launch_us": us, "achieved": flops / us / 1e6, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": flops / us / 1e6 / PEAK_BF16_TFLOPS, "bound": "mfma"})

  x = torch.randn(8, 64, 64, 320, device=dev).bfloat16()
  w = torch.randn(320, 320, 3, 3, device=dev) * 0.02
  timed(lambda: ops.conv3x3(x, w), 2.0 * 8 * 4096 * 320 * 2880, "gemm_kernel<8,160,1,0,3>: 3x3 conv 320->320 @ 64x64 x 8 (implicit GEMM, 256x160 ping-pong tile)")
  q = torch.randn(8, 4096, 320, device=dev).bfloat16()
  timed(lambda: ops.attention(q, q, q, 8), 4.0 * 8 * 8 * 4096 * 4096 * 40, "attention_dma_kernel<48,512>: self-attention N=4096, 8 heads x d=40, x 8")
  a = torch.randn(32768, 1280, device=dev).bfloat16()
  wl = (torch.randn(320, 1280, device=dev) * 0.03).bfloat16()
  timed(lambda: ops.gemm(a, wl), 2.0 * 32768 * 320 * 1280, "gemm_kernel<4,160,0,4,2>: GEMM 32768 x 320 x 1280 (level-0 feed-forward output shape)")
  return out


def cpu_baseline(n_infer_steps, pipe, unet_cfg, fp8):
  """The CPU oracle timed on this host's cores on a bounded sample of the same workload: ONE full-size UNet forward of a CFG
  pair (batch 2) ON THE WEIGHTS THE GPU HANDLE HOLDS + the GILLMapper, extrapolated to images/s (OPT-6.7b fp32 would need 27 GB
  and minutes; it is 0.65 % of an image's FLOPs and is left out of the sample, which flatters the CPU slightly).
  The oracle's output doubles as the bench's correctness gate: the same forward through gill_unet_forward must land within
  FWD_BAR of it (bench.py's own step-vs-step check only proves determinism).  Returns (record, forward_check)."""
  from gill_amd import synth
  from oracle import mapper_ref, unet_ref
  cores = synth.host_cores()
  torch.set_num_threads(cores)
  L, cd = unet_cfg.sample_size, unet_cfg.cross_attention_dim
  x = synth.initial_latents(2, 4, L)
  ctx = synth.normal("cpu_ctx", (2, unet_cfg.ctx_len, cd), 0).bfloat16().float()
  t = torch.tensor([961.0, 961.0])
  heads = unet_cfg.heads_per_level if unet_cfg.heads_per_level else unet_cfg.num_heads
  t0 = time.time()
  ref = unet_ref.unet_forward(pipe.bench_unet_sd_cpu, x, t, ctx, unet_cfg.block_out_channels, heads, unet_cfg.norm_num_groups)
  t_unet = time.time() - t0
  got = pipe.unet(x, t, ctx).float().cpu()
  rel = float(((got - ref).norm() / ref.norm()).item())
  cos = float(torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0).item())
  bar = 8e-2 if fp8 else 5e-2          # the bars of tests/test_stages_gpu.py (bf16) and tests/test_fp8_gpu.py (fp8 convolutions)
  check = {"what": f"one full-size UNet forward (batch 2, t = 961, {L}x{L} latents) through gill_unet_forward vs the fp32 CPU oracle on the same weights",
           "rel_l2": rel, "cosine": cos, "bar_rel_l2": bar, "ok": bool(rel == rel and rel < bar)}
  msd = synth.mapper_state_dict(synth.MapperConfig(in_dim=4096, out_dim=cd), seed=0)
  t0 = time.time()
  mapper_ref.mapper_forward(msd, synth.normal("cpu_x", (1, 8, 4096), 0), synth.normal("cpu_e", (1, 8, 4096), 0))
  t_map = time.time() - t0
  per_image = (n_infer_steps + 1) * t_unet + t_map
  rec = {"value": 1.0 / per_image, "unit": "images/s", "cores": cores, "kind": "port",
         "sample": f"UNet loop + GILLMapper only: 1 UNet forward of the CFG pair (batch 2, {L}x{L} latents): {t_unet:.2f} s; GILLMapper B=1: {t_map * 1e3:.0f} ms; "
                   f"extrapolated x{n_infer_steps + 1} UNet calls per image (OPT forward excluded from the sample).  Measured end to end once, not in this run: "
                   f"BASELINE configs[0] (opt-125m + SD-1.5, 1 prompt, 10 steps, fp32) through the oracle on 16 host cores = 54.4 s per image "
                   f"(profiles/r04_c1_cpu_end_to_end.log, tools/c1_cpu_end_to_end.py)"}
  return rec, check


class ClockSampler:
  """sclk (MHz) and socket power (W) of this rank's GPU, sampled every 0.25 s INSIDE the timed region by a host thread (boxes of
  the pool differ by up to 7 % at the same build; with these two keys in the line a slow box reads as a box, not as a regression).  Source: the amdgpu
  hwmon files of the card whose PCI address matches the torch device (freq1_input in Hz, power1_average / power1_input in uW: two file reads per
  sample, no subprocess); `rocm-smi --showclocks --showpower` once per second where sysfs has neither."""

  def __init__(self, dev):
    import glob
    import threading
    self.f_clk = self.f_pow = None
    self.clk, self.pow = [], []
    self._stop = threading.Event()
    self._thr = None
    self.source = None
    try:
      pr = torch.cuda.get_device_properties(dev)
      want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
    except Exception:
      want = None
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
    pick = [c for c in cards if want and want in os.path.realpath(c)] or [c for c in cards if glob.glob(c + "/hwmon/hwmon*/freq1_input")]
    for c in pick[:1]:
      for h in glob.glob(c + "/hwmon/hwmon*"):
        if os.path.exists(h + "/freq1_input"):
          self.f_clk = h + "/freq1_input"
        for name in ("power1_average", "power1_input"):
          if self.f_pow is None and os.path.exists(h + "/" + name):
            self.f_pow = h + "/" + name
    if self.f_clk or self.f_pow:
      self.source = "sysfs hwmon"
    else:
      import shutil
      self.smi = shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)
      self.dev_index = dev.index or 0
      self.source = "rocm-smi" if self.smi else None

  def _read(self, path, scale):
    try:
      with open(path) as fh:
        return float(fh.read().split()[0]) * scale
    except Exception:
      return None

  def _loop(self):
    import re
    while not self._stop.is_set():
      if self.source == "sysfs hwmon":
        c = self._read(self.f_clk, 1e-6) if self.f_clk else None
        p = self._read(self.f_pow, 1e-6) if self.f_pow else None
        period = 0.25
      else:
        c = p = None
        period = 1.0
        try:
          out = subprocess.run([self.smi, "-d", str(self.dev_index), "--showclocks", "--showpower"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                               timeout=5).stdout.decode(errors="replace")
          m = re.search(r"sclk clock level:\s*\d+:?\s*\((\d+)Mhz\)", out)
          c = float(m.group(1)) if m else None
          m = re.search(r"Power \(W\):\s*([0-9.]+)", out)
          p = float(m.group(1)) if m else None
        except Exception:
          pass
      if c:
        self.clk.append(c)
      if p:
        self.pow.append(p)
      self._stop.wait(period)

  def start(self):
    if self.source:
      import threading
      self._thr = threading.Thread(target=self._loop, daemon=True)
      self._thr.start()

  def stop(self):
    self._stop.set()
    if self._thr:
      self._thr.join(timeout=6)
    mean = lambda v: (sum(v) / len(v)) if v else None   # noqa: E731
    return {"sclk_mhz_mean": mean(self.clk), "sclk_mhz_min": min(self.clk) if self.clk else None, "power_w_mean": mean(self.pow),
            "samples": max(len(self.clk), len(self.pow)), "source": self.source}


def respawn_under_torchrun(a):
  """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
  with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
  env = dict(os.environ)
  env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
         "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
  sys.exit(subprocess.run(cmd, env=env).returncode)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=5)
  ap.add_argument("--warmup", type=int, default=2)
  ap.add_argument("--prompts-per-gpu", type=int, default=0, help="default: 4 at --gpus 1 (BASELINE configs[1]), 8 at --gpus > 1 (configs[2]), 16 with --config c5 (configs[4])")
  ap.add_argument("--total-prompts", type=int, default=0, help="global batch (default prompts-per-gpu x gpus); any value: uneven and empty shards are "
                  "legal (contiguous split, first B %% N ranks take one more)")
  ap.add_argument("--no-scale-origin", action="store_true", help="skip the 8-prompts-per-GPU side measurement of the N = 1 line")
  ap.add_argument("--infer-steps", type=int, default=50)
  ap.add_argument("--prompt-len", type=int, default=24)
  ap.add_argument("--small", action="store_true", help="opt-125m shapes (debug only; not the benchmark config)")
  ap.add_argument("--config", default="c2", choices=["c2", "c4", "c5"], help="c5: BASELINE configs[4], c2 with the UNet's resnet convolutions in fp8 e4m3 (not a parity mode); c2 (default): BASELINE configs[1]/[2], SD-1.5 UNet 512x512; "
                  "c4: configs[3], SD-2.1-768 UNet (1024-d context, head dim 64, v-prediction, 96x96 latents) + gen_emb_dim=1024 mapper")
  ap.add_argument("--fp8-convs-only", action="store_true", help="--config c5 without the fp8 GEGLU projections (the round-5 form of the mode, for A/B)")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc passes that measure roofline.traffic")
  ap.add_argument("--pmc-timeout", type=int, default=300)
  ap.add_argument("--child-pmc", action="store_true", help=argparse.SUPPRESS)   # the profiled child of pmc_traffic_live
  ap.add_argument("--backend", default="nccl", help="torch.distributed backend: nccl (= RCCL, the product path) | gloo (test rigs "
                  "where several ranks share one GPU; with --share-gpu every rank uses cuda:0)")
  ap.add_argument("--share-gpu", action="store_true", help=argparse.SUPPRESS)
  ap.add_argument("--scale-origin-value", type=float, default=0.0, help="images/s of the N = 1 line's scale_origin (8 prompts on one GPU): the "
                  "N > 1 line then carries efficiency_vs_scale_origin = value / (N x this)")
  a = ap.parse_args()

  if a.share_gpu:
    # test rig: several ranks on ONE GPU.  The UNet's in-kernel GroupNorm finishes wait inside a launch and need the device's CUs to themselves
    # (include/gill_amd.h "Exclusive-device contract"): two ranks' waiting launches side by side starve each other until the bounded waits give up.
    # The library reads the switch at its first launch.
    os.environ["GILL_GEMM_COOP"] = "0"
  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local = int(os.environ.get("LOCAL_RANK", "0"))
  if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
    respawn_under_torchrun(a)
  if world > 1:
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if a.share_gpu:
      local = 0
    if a.backend == "nccl":
      dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
      dist.init_process_group(a.backend, rank=rank, world_size=world)
  assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
  if a.prompts_per_gpu <= 0:
    a.prompts_per_gpu = 16 if a.config == "c5" else (4 if a.gpus == 1 else 8)   # configs[4]: batch 128 over 8 GPUs
  dev = torch.device("cuda", local)
  torch.cuda.set_device(dev)

  from gill_amd import synth
  opt_cfg = synth.OptConfig.opt_125m() if a.small else synth.OptConfig.opt_6_7b()
  unet_cfg = synth.UNetConfig.sd21_768() if a.config == "c4" else synth.UNetConfig.sd15()
  if a.config == "c5":
    unet_cfg.fp8_convs = 2 if a.fp8_convs_only else True       # (gill_unet_config.fp8_convs: 1 = convolutions + GEGLU projections, 2 = convolutions only)
  vae_cfg = synth.VAEConfig(latent_size=unet_cfg.sample_size)
  tflop_fwd = UNET_TFLOP_SD21_768 if a.config == "c4" else UNET_TFLOP_PER_SAMPLE_FORWARD
  side = 8 * unet_cfg.sample_size
  P = a.prompts_per_gpu
  from gill_amd.parallel import shard_bounds
  B_total = a.total_prompts if a.total_prompts > 0 else P * world
  lo, hi = shard_bounds(B_total, rank, world)
  P_local = hi - lo                        # == P unless --total-prompts makes the shards uneven
  P_max = (B_total + world - 1) // world
  # the N = 1 headline line also carries the 8-prompts-per-GPU figure (the per-GPU load of the N > 1 lines): the origin of the
  # driver's 1 -> 8 GPU weak-scaling curve must be measured at the same per-GPU work as its other points
  want_origin = (world == 1 and a.config == "c2" and not a.small and not a.no_scale_origin and not a.child_pmc and
                 a.total_prompts == 0 and P != 8)
  t_build0 = time.time()
  want_cpu = world == 1 and not a.no_cpu_baseline and not a.child_pmc and not a.small
  g = build_model(dev, opt_cfg, unet_cfg, max(P_max, 8 if want_origin else 1), vae_cfg, keep_unet_cpu=want_cpu)
  torch.cuda.synchronize()
  t_build = time.time() - t_build0
  ids = synth.synthetic_prompt_ids(B_total, a.prompt_len, seed=0)[:, :a.prompt_len]   # [IMG] ids are appended by generate_images
  lat0 = synth.initial_latents(B_total, 4, unet_cfg.sample_size, seed=1337).to(dev)

  # HIP events around the UNet loop: recorded on torch's current stream, which gill_sd_denoise fences its private launch stream
  # to on both sides (event record -> stream wait), so the pair brackets exactly the loop's kernels
  ev = {"t": []}
  orig_call = g.sd_pipe.__class__.__call__

  def timed_call(self, *args, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig_call(self, *args, **kw)
    e1.record()
    ev["t"].append((e0, e1))
    return out
  g.sd_pipe.__class__.__call__ = timed_call

  # stages 1 and 2 the same way (their launches go to torch's current stream): OPT forward -> [IMG] hidden states, GILLMapper
  opt_ev, map_ev = [], []

  def wrap_events(obj, name, store):
    fn = getattr(obj, name)

    def timed(*args, **kw):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      o = fn(*args, **kw)
      e1.record()
      store.append((e0, e1))
      return o
    setattr(obj, name, timed)
  wrap_events(g.model, "img_hidden_states", opt_ev)
  wrap_events(g.model.gen_text_hidden_fcs[0], "forward", map_ev)

  vae_ev = []
  orig_dec = g.sd_pipe.__class__.decode_latents

  def timed_dec(self, *args, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    o = orig_dec(self, *args, **kw)
    e1.record()
    vae_ev.append((e0, e1))
    return o
  g.sd_pipe.__class__.decode_latents = timed_dec

  def step():   # prompts -> OPT -> mapper -> 51 UNet calls -> latents (all-gathered) -> VAE decode -> uint8 512x512 images
    return g.generate_images(ids, num_inference_steps=a.infer_steps, guidance_scale=7.5, latents=lat0, decode=True)

  for _ in range(a.warmup):
    step()
  ev["t"].clear()
  vae_ev.clear()
  opt_ev.clear()
  map_ev.clear()
  kept = []
  sampler = ClockSampler(dev) if (rank == 0 and not a.child_pmc) else None
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  if sampler:
    sampler.start()
  t0 = time.perf_counter()
  for _ in range(a.steps):
    kept.append(step())      # (latents of all ranks, uint8 images of this rank): 0.25 + 3 MiB per prompt, checked below
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  clocks = sampler.stop() if sampler else None
  from gill_amd import _native as _N
  n_giveups = _N.lib().gill_coop_timeouts()
  if n_giveups != 0:
    print(f"[bench rank {rank}] {n_giveups} in-kernel GroupNorm finish(es) timed out (GPU shared with another waiting launch?): outputs are NaN-poisoned; "
          "rerun with GILL_GEMM_COOP=0", file=sys.stderr)
    sys.exit(5)
  per_rank = None
  if world > 1:
    # every rank's own wall time and model-build time (a straggler or a slow loader shows up in the line), then the MAX as the job's time
    mine = torch.tensor([dt, t_build], device=dev, dtype=torch.float64)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    per_rank = {"ms_per_step": [float(t[0].item()) / a.steps * 1e3 for t in allr], "model_build_s": [round(float(t[1].item()), 1) for t in allr]}
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
  if a.child_pmc:
    return

  # every timed step must have produced the same finite result (same inputs each step; no atomics on the path)
  ref_lat, ref_img = kept[0][0].float(), kept[0][1].float()
  table, ok = [], True
  for i, (lat, img) in enumerate(kept):
    shape_ok = tuple(lat.shape) == (B_total, 4, unet_cfg.sample_size, unet_cfg.sample_size) and \
        tuple(img.shape) == (P_local, side, side, 3) and img.dtype == torch.uint8
    lat = lat.float()
    n_bad = int((~torch.isfinite(lat)).sum().item())
    rel = float(((lat - ref_lat).norm() / ref_lat.norm()).item()) if n_bad == 0 else float("nan")
    img_mad = float((img.float() - ref_img).abs().mean().item())
    img_std = float(img.float().std().item()) if P_local > 0 else float("nan")
    good = shape_ok and n_bad == 0 and rel <= REL_STEP_TOL and (P_local == 0 or img_std > 1.0)
    ok &= good
    table.append((i, shape_ok, n_bad, rel, img_mad, img_std, good))
  okt = torch.tensor([1 if ok else 0], device=dev)
  if world > 1:
    dist.all_reduce(okt, op=dist.ReduceOp.MIN)
  if not ok:
    print(f"[bench rank {rank}] OUTPUT CHECK FAILED (tolerance rel-L2 {REL_STEP_TOL} vs the first timed step):", file=sys.stderr)
    for row in table:
      print("  step %3d shape_ok=%s nonfinite=%d rel_l2_vs_step0=%.3e image_mean_abs_diff=%.3f image_std=%.2f %s" %
            (row[:6] + ("ok" if row[6] else "BAD",)), file=sys.stderr)
  if int(okt.item()) == 0:
    if world > 1:
      dist.destroy_process_group()
    sys.exit(3)
  max_rel = max(r[3] for r in table)

  scale_origin = None
  if want_origin and rank == 0:
    # same model, same loop, 8 prompts on the GPU (UNet batch 16): BASELINE configs[2]'s per-GPU share
    ids8 = synth.synthetic_prompt_ids(8, a.prompt_len, seed=0)[:, :a.prompt_len]
    lat8 = synth.initial_latents(8, 4, unet_cfg.sample_size, seed=1337).to(dev)

    def step8():
      return g.generate_images(ids8, num_inference_steps=a.infer_steps, guidance_scale=7.5, latents=lat8, decode=True)
    ev_main, vae_main, opt_main, map_main = list(ev["t"]), list(vae_ev), list(opt_ev), list(map_ev)
    first = step8()
    ev["t"].clear()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    n8 = max(2, min(a.steps, 4))
    outs8 = [step8() for _ in range(n8)]
    torch.cuda.synchronize()
    dt8 = time.perf_counter() - t1
    ms8 = sum(e0.elapsed_time(e1) for e0, e1 in ev["t"]) / max(1, len(ev["t"]))
    same8 = all(torch.equal(o[0], first[0]) for o in outs8) and bool(torch.isfinite(first[0]).all().item())
    ach8 = tflop_fwd * 2 * (a.infer_steps + 1) * 8 / (ms8 * 1e-3)
    scale_origin = {"prompts_per_gpu": 8, "value": 8 * n8 / dt8, "unit": "images/s", "steps": n8, "ms_per_step": dt8 / n8 * 1e3,
                    "frac": ach8 / PEAK_BF16_TFLOPS, "achieved": ach8, "avg_launch_ms": ms8, "all_steps_bit_identical": same8,
                    "note": "N = 1 at the per-GPU load of the N > 1 lines (BASELINE configs[2]: 8 prompts per GPU): like-for-like origin of the weak-scaling curve"}
    if not same8:
      print("[bench] scale_origin: 8-prompt steps differ or are non-finite", file=sys.stderr)
      sys.exit(3)
    ev["t"][:] = ev_main
    vae_ev[:] = vae_main
    opt_ev[:] = opt_main
    map_ev[:] = map_main
    del outs8, first

  if rank == 0:
    images = B_total * a.steps
    value = images / dt
    unet_ms = sum(e0.elapsed_time(e1) for e0, e1 in ev["t"]) / max(1, len(ev["t"]))   # per sd_pipe call (<= 8 prompts)
    vae_ms = sum(e0.elapsed_time(e1) for e0, e1 in vae_ev) / max(1, len(vae_ev))
    per_call = min(P_local, 8)                                                         # gen_max_bs = 8 chunks (models.py:726)
    if P_local > 8 and P_local % 8:
      per_call = P_local / ((P_local + 7) // 8)                                        # (average over the chunks of this rank)
    flop_per_call = tflop_fwd * 2 * (a.infer_steps + 1) * per_call
    achieved = flop_per_call / (unet_ms * 1e-3)
    traffic, traffic_detail = (None, "skipped (--no-pmc or N > 1)")
    if world == 1 and not a.no_pmc and not a.small and a.config == "c2":
      del kept
      traffic, traffic_detail = pmc_traffic_live(a)
    cfg_name = "configs[1]" if (world == 1 and P == 4) else ("configs[2]" if P == 8 else "custom")
    if a.total_prompts > 0:
      cfg_name = "custom (--total-prompts)"
    unet_name = "SD-1.5 UNet"
    metric = "512x512 images/sec/node, OPT-6.7B+SD1.5 50-step"
    if a.config == "c4":
      cfg_name, unet_name = "configs[3]", "SD-2.1-768 UNet (outside the reference: needs the gen_emb_dim=1024 mapper variant)"
      metric = "768x768 images/sec/node, OPT-6.7B+SD2.1-768 50-step (BASELINE configs[3]; not the headline metric)"
    dtype = "bf16"
    if a.config == "c5":
      lin = "" if a.fp8_convs_only else " and the 11 GEGLU projections of levels 1-3"
      cfg_name, unet_name = "configs[4]", f"SD-1.5 UNet with fp8 (e4m3) resnet convolutions{lin}, everything else bf16"
      metric = "512x512 images/sec/node, OPT-6.7B+SD1.5 50-step, fp8 resnet convolutions + GEGLU projections (BASELINE configs[4]; not the headline metric)"
      dtype = "fp8 e4m3 (44 resnet convolutions" + ("" if a.fp8_convs_only else " + 11 GEGLU projections") + ") + bf16"
    rec = {
      "metric": metric, "value": value, "unit": "images/s",
      "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
      "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
      "config": {"workload": f"BASELINE {cfg_name}: {'opt-125m' if a.small else 'opt-6.7b'} + GILLMapper + {unet_name} + VAE decoder "
                             f"(random-init weights of the exact shapes), {P if a.total_prompts == 0 else 'uneven'} prompts/GPU x {world} GPU = batch {B_total}, prompt "
                             f"{a.prompt_len}+8 [IMG] tokens, {a.infer_steps} PLMS steps ({a.infer_steps + 1} UNet calls, CFG 7.5, UNet batch "
                             f"{2 * per_call}), final latents all-gathered, then VAE decode of the local shard to uint8 {side}x{side} "
                             f"({vae_ms:.1f} ms per {P_local} images)", "parallelism": f"dp{world}", "prompts_per_gpu": P if a.total_prompts == 0 else B_total / world,
                 "global_batch": B_total, "model_build_s": round(t_build, 1),
                 "collective": (f"{a.backend} ({'RCCL ' + '.'.join(str(v) for v in torch.cuda.nccl.version()) if a.backend == 'nccl' else 'test rig'}): one all_gather_into_tensor of the final latents per step"
                                if world > 1 else "none (single rank)")},
      "output_check": {"steps_checked": len(table), "max_rel_l2_vs_first_step": max_rel, "tolerance": REL_STEP_TOL, "all_finite": True},
      "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                   "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic,
                   "traffic_unit": "GB of HBM traffic per step (all hot-path kernels of the step; live rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE passes)",
                   "traffic_detail": traffic_detail,
                   "kernel": f"{unet_name.split(' (')[0]} denoise loop (gill_sd_denoise: MFMA GEMM/implicit-conv + flash attention kernels)",
                   "algorithmic_tflop_per_launch": flop_per_call, "avg_launch_ms": unet_ms},
    }
    # every stage's own time and roofline (HIP events on the launch stream, averaged over the timed steps; SURVEY.md section 8d: the OPT
    # pass is bound by streaming its decoder weights from HBM once per batch, the mapper by launch latency, the UNet loop by MFMA)
    avg = lambda evs: sum(e0.elapsed_time(e1) for e0, e1 in evs) / max(1, len(evs))   # noqa: E731
    opt_ms, map_ms = avg(opt_ev), avg(map_ev)
    D, F, NL = opt_cfg.hidden_size, opt_cfg.ffn_dim, opt_cfg.num_layers
    opt_gb = 2.0 * NL * (4 * D * D + 2 * D * F) / 1e9        # bf16 decoder matrices, each read once per batch
    n_sd_calls = max(1, (P_local + 7) // 8)
    rec["stages"] = {
      "opt": {"ms": opt_ms, "GB": opt_gb, "TBps": opt_gb / max(opt_ms, 1e-9), "frac_of_8TBps": opt_gb / max(opt_ms, 1e-9) / 8.0, "bound": "hbm",
              "what": f"gill_opt_img_hidden: {NL}-layer OPT forward of {P_local} x {a.prompt_len + 8} tokens, decoder weights streamed once"},
      "mapper": {"us": map_ms * 1e3, "bound": "latency", "what": "gill_mapper_forward (4 + 4 transformer layers on 8 -> 77 tokens per prompt)"},
      "unet_loop": {"ms": unet_ms * n_sd_calls, "frac_of_2.5PFLOPs": achieved / PEAK_BF16_TFLOPS, "bound": "mfma",
                    "what": f"gill_sd_denoise x {n_sd_calls}: {a.infer_steps + 1} UNet forwards of batch {2 * per_call}"},
      "vae": {"ms": vae_ms, "what": f"gill_vae_decode of {P_local} latents to uint8 {side}x{side}"},
      "host_and_gaps_ms": dt / a.steps * 1e3 - (opt_ms + map_ms + unet_ms * n_sd_calls + vae_ms),
    }
    # the box under this very timed region (sampled by a host thread between t0 and dt): the two keys VERDICT r05 item 5 asks for, top level
    rec["sclk_mhz_mean"] = clocks["sclk_mhz_mean"] if clocks else None
    rec["power_w_mean"] = clocks["power_w_mean"] if clocks else None
    rec["clocks"] = clocks
    if want_origin:
      rec["scale_origin"] = scale_origin
    if world == 1 and not a.small and a.config == "c2":
      rec["roofline_kernels"] = kernel_rooflines(dev)
    if per_rank is not None:
      rec["per_rank"] = per_rank
    if a.scale_origin_value > 0:
      rec["efficiency_vs_scale_origin"] = value / (world * a.scale_origin_value)
    rec["cpu_baseline"], rec["forward_check"] = None, None
    if want_cpu:
      # (every config: the oracle forward is both the reported CPU baseline and the correctness gate of this very build of the
      # UNet — c4 / c5 included, whose step-vs-step check alone could not see a deterministic error)
      rec["cpu_baseline"], rec["forward_check"] = cpu_baseline(a.infer_steps, g.sd_pipe, unet_cfg, a.config == "c5")
    print(json.dumps(rec), flush=True)
    if rec["forward_check"] is not None and not rec["forward_check"]["ok"]:
      print(f"[bench] FORWARD CHECK FAILED: {rec['forward_check']}", file=sys.stderr)
      sys.exit(4)
  if world > 1:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()

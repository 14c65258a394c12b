#!/usr/bin/env python3
"""The OPT-6.7b stage alone (bench.py's model, gill_opt_img_hidden on 4 or 8 prompts of 24 + 8 tokens), N calls: for rocprofv3 kernel traces of
the stage (tools/sessions/r05_s07.sh) and for its HIP-event time without the rest of a bench step.
  python tools/opt_only.py [prompts] [calls]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gill_amd
gill_amd.configure_hip_runtime()
import bench
from types import SimpleNamespace
from gill_amd import synth
from gill_amd.models import GILL

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
ocfg = synth.OptConfig.opt_6_7b()
osd = bench.gpu_state_dict(lambda c, meta: bench.shapes_of("opt_state_dict", c), ocfg, dev, 0)
args = SimpleNamespace(freeze_lm=True, freeze_vm=True, opt_version="facebook/opt-6.7b", visual_encoder="openai/clip-vit-large-patch14",
                       n_visual_tokens=4, ret_emb_dim=256, gen_emb_dim=768, text_emb_layers=[-1], text_fc_mode="gill_mapper",
                       ret_text_fc_mode="linear", num_tokens=8, num_clip_tokens=77, retrieval_token_idx=synth.IMG_TOKEN_IDS,
                       gen_token_idx=synth.IMG_TOKEN_IDS, opt_state_dict=osd)
g = GILL(synth.HashTokenizer(), args, load_sd=False).eval().bfloat16().cuda()
del osd
ids = synth.synthetic_prompt_ids(P, 24, seed=1)[:, :24]
full = torch.cat([ids, torch.tensor(synth.IMG_TOKEN_IDS)[None].expand(P, -1)], 1).to(dev)
last = torch.full((P,), 31)
for _ in range(3):
  g.model.img_hidden_states(full, last)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
  g.model.img_hidden_states(full, last)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(f"OPT-6.7b img_hidden, {P} prompts x 32 tokens: {ms:.3f} ms per call = {12.884901888 / ms:.2f} TB/s of decoder weights")

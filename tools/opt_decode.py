#!/usr/bin/env python3
"""KV-cached greedy decoding on the OPT-6.7b shapes (GILLModel.generate, batch 1, 8-token prompt, N new tokens): ms per decoded token.
  python tools/opt_decode.py [new_tokens]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda:0")
ocfg = synth.OptConfig.opt_6_7b()
osd = bench.gpu_state_dict(lambda c, meta: bench.shapes_of("opt_state_dict", c), ocfg, dev, 0)
args = SimpleNamespace(freeze_lm=True, freeze_vm=True, opt_version="facebook/opt-6.7b", visual_encoder="openai/clip-vit-large-patch14",
                       n_visual_tokens=4, ret_emb_dim=256, gen_emb_dim=768, text_emb_layers=[-1], text_fc_mode="gill_mapper",
                       ret_text_fc_mode="linear", num_tokens=8, num_clip_tokens=77, retrieval_token_idx=synth.IMG_TOKEN_IDS,
                       gen_token_idx=synth.IMG_TOKEN_IDS, opt_state_dict=osd)
g = GILL(synth.HashTokenizer(), args, load_sd=False).eval().bfloat16().cuda()
del osd
ids = synth.synthetic_prompt_ids(1, 8, seed=1)[:, :8].to(dev)
emb = g.model.input_embeddings(ids)
kw = dict(min_word_tokens=n, temperature=0.0)          # [IMG] suppressed: n ordinary greedy steps
g.model.generate(emb, 4, use_kv_cache=True, **kw)
torch.cuda.synchronize()
t0 = time.time()
out, _, _ = g.model.generate(emb, n, use_kv_cache=True, **kw)
torch.cuda.synchronize()
dt = time.time() - t0
print(f"OPT-6.7b KV-cached decode, batch 1: {n} tokens in {dt * 1e3:.1f} ms = {dt / n * 1e3:.2f} ms per token ({12.88 / (dt / n * 1e3):.2f} TB/s of decoder weights + the 412 MB lm_head)")

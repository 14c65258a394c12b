"""CPU oracle of the GILL image-generation hot path — TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; nothing
under gill_amd/ does.  It restates, in plain fp32 torch-CPU ops, the algorithm the reference executes:

  stage 1  oracle.opt_ref      transformers OPTForCausalLM as called at gill/models.py:363-365, :465
  stage 2  oracle.mapper_ref   gill/layers.py:28-53 (TextFcLayer 'gill_mapper' = nn.Transformer, norm_first)
  stage 3  oracle.unet_ref     diffusers UNet2DConditionModel (SD-1.5 config) — NOT in the reference tree
           oracle.scheduler_ref  diffusers PNDMScheduler (skip_prk_steps) — NOT in the reference tree
           oracle.pipeline_ref   the driver loop gill/custom_sd.py:607-651 and the glue gill/models.py:164-441
           oracle.vae_ref        diffusers AutoencoderKL.decode + uint8 conversion (custom_sd.py:385-392) — NOT in the tree

  image prompts  oracle.clip_ref   transformers CLIPVisionModel + visual_embeddings as called at gill/models.py:129-146

Pinning status
  * stages 1-2 and the image-prompt branch are PINNED: tests/golden/*.npz hold outputs of the reference's own code
    (gill.layers.TextFcLayer, gill.models.GILLModel.forward/generate, GILL.generate_for_images_and_texts)
    imported from /root/reference by oracle/gen_golden.py, and tests/test_oracle_golden.py checks this
    restatement against them.
  * stage 3 is PARITY UNPINNED: diffusers==0.17.1 is a pinned, un-vendored dependency (requirements.txt:9)
    that is not installed here and cannot be fetched; the reference holds no test or golden vector for it.
    unet_ref / scheduler_ref / vae_ref restate the published SD-1.5 UNet2DConditionModel / PNDMScheduler / AutoencoderKL decoder algorithm and
    are anchored on the reference's call sites (gill/custom_sd.py:567-651, gill/models.py:724-731).
"""

"""The device half of the loop-parity tests: the driver loop of a loop fixture (tests/golden/G18..G30: the REFERENCE's own
text2image_ldm_stable recorded on CPU in fp32) run through the HIP path on the same seeded weights and trajectory.
Shared by tests/test_end_to_end.py and tools/loop_error_budget.py.  Nothing here touches oracle/."""
from __future__ import annotations

import torch

import cases
from _util import fixture_mismatch, load

# kind -> (fixture, cases.<config>, controller kind, full width, model name)
SD21, SD14, SDXL = "stabilityai/stable-diffusion-2-1-base", "CompVis/stable-diffusion-v1-4", "stabilityai/stable-diffusion-xl-base-1.0"
LOOP_KINDS = {
    "geometry_editor": ("G18_loop", "LOOP", "geometry_editor", False, SD21),
    "geometry_remover": ("G19_loop_remover", "LOOP", "geometry_remover", False, SD21),
    "cfg0": ("G20_loop_cfg0", "LOOP_CFG0", "geometry_editor", False, SD21),            # BASELINE configs[0]: 256^2, 2-D translation, 20 steps
    "sd14": ("G23_loop_sd14", "LOOP", "geometry_editor", False, SD14),                 # the reference's default model layout (head dims 40 / 80 / 160)
    "sdxl": ("G27_loop_sdxl", "LOOP_SDXL", "geometry_editor", False, SDXL),            # SDXL-base topology (narrow), 512^2
    "cfg0_full": ("G21_loop_cfg0_full", "LOOP_CFG0", "geometry_editor", True, SD21),   # ... at the FULL SD2.1-base width (865 M parameters)
    "cfg1_full": ("G22_loop_cfg1_full", "LOOP_CFG1", "geometry_editor", True, SD21),   # configs[1]'s shape: 512^2, 3-D rotation, 4 steps
    "remover_full": ("G26_loop_remover_full", "LOOP", "geometry_remover", True, SD21),
    "cfg1_t50": ("G28_loop_cfg1_t50", "LOOP_CFG1_T50", "geometry_editor", False, SD21),               # configs[1] at its stated length, narrow
    "rem768_t75": ("G29_loop_remover768_t75", "LOOP_REM768_T75", "geometry_remover", False, SD21),    # configs[3]'s length, narrow
    "cfg1_full_t50": ("G30_loop_cfg1_full_t50", "LOOP_CFG1_T50", "geometry_editor", True, SD21),      # configs[1] ITSELF: full width x 50 steps
}

_MODELS = {}


def cached_model(name, tiny, dtype):
    """The full-width models take ~10 s to build: one instance per (architecture, width, dtype) for the loop tests (they restore the
    processor and set the scheduler's timesteps themselves)."""
    from geodiffuser_amd.diffusion import load_model
    key = (name, tiny, dtype)
    if key not in _MODELS:
        if not tiny:
            for k in [k for k in _MODELS if not k[1]]:          # keep at most one 865 M-parameter model resident
                del _MODELS[k]
            torch.cuda.empty_cache()
        _MODELS[key] = load_model(name, device="cuda:0", tiny=tiny, dtype=dtype)
    return _MODELS[key]


def make_controller(kind, c, inp, device="cuda:0"):
    """The controller and loss weights of the loop fixtures (oracle/gen_golden.py run_reference_loop builds the reference's the same way)."""
    from geodiffuser_amd.attention_processors import AttentionGeometryEdit, AttentionGeometryRemover
    from geodiffuser_amd.generic_torch import torch_erode
    if kind == "geometry_editor":
        lw = {"self": {"sim": 55, "movement": 30.5, "removal": 2.6, "smoothness": 30.0, "amodal": 80.5},
              "cross": {"sim": 45, "movement": 30.34, "removal": 2.6, "smoothness": 15.0, "amodal": 3.5}}
        ctrl = AttentionGeometryEdit(["", ""], c["steps"], {"default_": c["cross_replace"]}, c["self_replace"], image_mask=inp["mask"],
                                     obj_edit_step=c["obj_edit_step"], device=device)
        ctrl.amodal_mask = torch_erode(torch.from_numpy(cases.amodal_input(inp["mask"], *c.get("amodal_shift", (32, -12)))))
    else:
        lw = {"self": {"sim": 55, "removal": 4.6, "smoothness": 30.0}, "cross": {"sim": 45, "removal": 4.6, "smoothness": 15.0}}
        ctrl = AttentionGeometryRemover(["", ""], c["steps"], {"default_": 0.9}, 0.9, image_mask=inp["mask"], obj_edit_step=1.0, device=device)
    ctrl.default_loss_weights = lw
    ctrl.initialize_default_loss_weights()
    return ctrl, lw


def run_device_loop(kind_name, dtype, skip_refs=(False, "3rows", True, "carry", "ahead")):
    """-> (fixture dict, fixture name, runs) with runs = [(latents f32 cpu, loss log, final removal weight, first latent update, weight
    trajectory, CFG passes that took their reference row from another pass, optimisation passes that ran on the edit row alone)] — one run
    per entry of ``skip_refs``: False = the reference's 4-row CFG batch; "3rows" = without the unused uncond_ref row; True = that, and at
    steps with an optimisation pass the reference row's layer tensors come from that pass (editor.REF_FROM_OPT: 2 rows); "carry" (r06): the
    reference row of an optimisation step additionally goes through the UNet one step ahead, carried by the previous step's CFG pass, and the
    optimisation pass runs forward + backward on the edit row alone (editor.REF_AHEAD = 2); "ahead" = the product default: ALL reference
    rows of the edit in one batched pass before the loop, every optimisation pass on the edit row alone (editor.REF_AHEAD = 1)."""
    from geodiffuser_amd import editor
    from geodiffuser_amd.attention_processors import VanillaAttentionProcessor
    fixture, cfgname, kind, full, name = LOOP_KINDS[kind_name]
    g = load(fixture)
    c = getattr(cases, cfgname)
    p, tok, sched = cached_model(name, not full, dtype)
    if name == SDXL:                     # the fixture's model was built for 512^2 micro-conditioning ids
        p.unet.default_added_cond = (p.unet.default_added_cond[0], torch.tensor([[512, 512, 0, 0, 512, 512]], dtype=torch.float32, device="cuda"))
    probe = torch.cat([q.detach().float().reshape(-1)[:64] for q in p.unet.parameters()]).cpu()
    if not torch.allclose(probe, torch.from_numpy(g["weight_probe"]), atol=2e-3 if dtype == torch.float16 else 2e-2):
        fixture_mismatch("seeded weights differ from the fixture's (different torch build): the fixture does not apply")
    inp = cases.loop_inputs(c)
    coords = torch.from_numpy(inp["coords"])
    ctrl, lw = make_controller(kind, c, inp)
    prev = (editor.NUM_DDIM_STEPS, editor.GUIDANCE_SCALE, editor.SKIP_OPTIM_STEPS, editor.SKIP_UNCOND_REF)
    prev_rfo, prev_ahead = editor.REF_FROM_OPT, editor.REF_AHEAD
    editor.NUM_DDIM_STEPS, editor.GUIDANCE_SCALE, editor.SKIP_OPTIM_STEPS = c["steps"], c["guidance"], c["skip_optim"]
    runs, updates, weights = [], [], []
    orig_apply = editor._apply_latent_update

    def rec_apply(latents_in, g_lat, context_in, g_ctx, l_eff, mask):
        res = orig_apply(latents_in, g_lat, context_in, g_ctx, l_eff, mask)
        if not updates:                                     # (the first one is compared; keeping all 17-32 would only hold memory)
            updates.append((res[0][-1:].detach().float() - latents_in[-1:].detach().float()).cpu())
        weights.append(float(ctrl.loss_weight_dict["self"]["removal"]))      # the adaptive weight in effect AT this pass
        return res

    editor._apply_latent_update = rec_apply
    try:
        for skip_ref in skip_refs:
            editor.SKIP_UNCOND_REF = bool(skip_ref)
            editor.REF_FROM_OPT = prev_rfo and skip_ref in (True, "carry", "ahead")
            editor.REF_AHEAD = 0 if not prev_ahead else {"carry": 2, "ahead": 1}.get(skip_ref, 0)
            n_rfo, n_ahead = editor.REF_FROM_OPT_PASSES, editor.REF_AHEAD_PASSES
            updates.clear()
            weights.clear()
            ctrl.reset() if hasattr(ctrl, "reset") else None
            ctrl.masks_cache_dict = {}
            ctrl.default_loss_weights = {k: dict(v) for k, v in lw.items()}
            ctrl.initialize_default_loss_weights()
            ddim = [torch.from_numpy(a).to("cuda").to(dtype) for a in inp["ddim_latents"]]
            lat, _, log = editor.text2image_ldm_stable(
                p, ["", ""], ctrl, latent=torch.from_numpy(inp["x_T"]).to("cuda").to(dtype), num_inference_steps=c["steps"],
                guidance_scale=c["guidance"], uncond_embeddings=None, transform_coordinates=coords, mask_obj=torch.from_numpy(inp["mask"]),
                optimize_steps=c["optimize_steps"], latent_replace=c["latent_replace"], lr=c["lr"], optimize_embeddings=True,
                optimize_latents=True, ddim_latents=ddim, ddim_noise=None, edit_type=kind, fast_start_steps=0.0,
                num_first_optim_steps=1, use_adaptive_optimization=True, return_type="latents", image_size=c["size"])
            runs.append((lat.float().cpu(), log, float(ctrl.loss_weight_dict["self"]["removal"]), updates[0].clone(), list(weights),
                         editor.REF_FROM_OPT_PASSES - n_rfo, editor.REF_AHEAD_PASSES - n_ahead))
    finally:
        editor._apply_latent_update = orig_apply
        editor.REF_FROM_OPT, editor.REF_AHEAD = prev_rfo, prev_ahead
        editor.NUM_DDIM_STEPS, editor.GUIDANCE_SCALE, editor.SKIP_OPTIM_STEPS, editor.SKIP_UNCOND_REF = prev
        p.unet.set_attn_processor(VanillaAttentionProcessor())
    return g, fixture, runs


def run_device_loop_batch(kind_name, dtype, extra_seeds=(5,)):
    """The fixture's edit as edit 0 of an in-process BATCH (geodiffuser_amd.batch.text2image_ldm_stable_batch) next to ``len(extra_seeds)``
    other edits of the same configuration (other masks, transforms and trajectories: cases.loop_inputs with another seed / ellipse).
    -> (fixture dict, fixture name, (latents [2,4,h,w] f32 cpu of edit 0, its loss log, its final removal weight, its weight trajectory),
    [(latents, log) of the other edits])."""
    import numpy as np
    from geodiffuser_amd import batch as GB, editor
    from geodiffuser_amd.attention_processors import VanillaAttentionProcessor
    fixture, cfgname, kind, full, name = LOOP_KINDS[kind_name]
    g = load(fixture)
    c = getattr(cases, cfgname)
    p, tok, sched = cached_model(name, not full, dtype)
    probe = torch.cat([q.detach().float().reshape(-1)[:64] for q in p.unet.parameters()]).cpu()
    if not torch.allclose(probe, torch.from_numpy(g["weight_probe"]), atol=2e-3 if dtype == torch.float16 else 2e-2):
        fixture_mismatch("seeded weights differ from the fixture's (different torch build): the fixture does not apply")
    cfgs = [c] + [dict(c, seed=c["seed"] + 1000 * s_, transform=None if c.get("transform") == "rotate" else c.get("transform"),
                       ellipse=dict(cx=0.42 * c["size"] + 7 * s_, cy=0.55 * c["size"] - 5 * s_, ax=0.13 * c["size"], ay=0.1 * c["size"]))
                  for s_ in extra_seeds]
    inps = [cases.loop_inputs(cc) for cc in cfgs]
    subs, coords = [], []
    for cc, inp in zip(cfgs, inps):
        ctrl, lw = make_controller(kind, cc, inp)
        ctrl.default_loss_weights = {k: dict(v) for k, v in lw.items()}
        ctrl.initialize_default_loss_weights()
        subs.append(ctrl); coords.append(torch.from_numpy(inp["coords"]))
    B = len(subs)
    batch = GB.EditBatch(subs, coords)
    prev = (editor.NUM_DDIM_STEPS, editor.GUIDANCE_SCALE, editor.SKIP_OPTIM_STEPS)
    editor.NUM_DDIM_STEPS, editor.GUIDANCE_SCALE, editor.SKIP_OPTIM_STEPS = c["steps"], c["guidance"], c["skip_optim"]
    weights = []
    orig = GB.ops.masked_latent_update
    calls = [0]

    def rec(x, gr, m, step):                                  # edit 0's adaptive weight in effect AT each optimisation pass
        if calls[0] % B == 0:
            weights.append(float(subs[0].loss_weight_dict["self"]["removal"]))
        calls[0] += 1
        return orig(x, gr, m, step)

    GB.ops.masked_latent_update = rec
    try:
        T = c["steps"]
        ddim = [torch.cat([torch.from_numpy(inp["ddim_latents"][t]) for inp in inps]).to("cuda").to(dtype) for t in range(T + 1)]
        x_T = torch.cat([torch.from_numpy(inp["x_T"]) for inp in inps]).to("cuda").to(dtype)
        lat, logs = GB.text2image_ldm_stable_batch(
            p, [""] * B, batch, c["steps"], c["guidance"], latent=x_T, ddim_latents=ddim, masks_obj=[torch.from_numpy(inp["mask"]) for inp in inps],
            optimize_steps=c["optimize_steps"], latent_replace=c["latent_replace"], lr=c["lr"], optimize_embeddings=True, optimize_latents=True,
            edit_type=kind, use_adaptive_optimization=True, return_type="latents", image_size=c["size"], skip_optim_steps=c["skip_optim"],
            num_ddim_steps=c["steps"])
    finally:
        GB.ops.masked_latent_update = orig
        editor.NUM_DDIM_STEPS, editor.GUIDANCE_SCALE, editor.SKIP_OPTIM_STEPS = prev
        p.unet.set_attn_processor(VanillaAttentionProcessor())
    lat = lat.float().cpu()
    first = (torch.stack([lat[0], lat[B]]), logs[0], float(subs[0].loss_weight_dict["self"]["removal"]), list(weights))
    others = [(torch.stack([lat[j], lat[B + j]]), logs[j]) for j in range(1, B)]
    return g, fixture, first, others

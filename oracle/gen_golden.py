"""ORACLE tooling (build container only): record golden vectors from the REFERENCE's own Python.

    python oracle/gen_golden.py            # writes tests/golden/*.npz

Inputs are regenerated from seeds by ``tests/golden/cases.py``; the fixtures hold only what the
reference computed for them (plus bit-packed masks).  The reference is imported under the stand-in
modules of ``oracle/ref_import.py``; nothing of it is copied or shipped.

Fixture ids follow SURVEY.md 8c (G1..G13).
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases  # noqa: E402
import ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def provenance():
    """Where a fixture was recorded: long loops are reproducible to the bit only on the same torch build / thread count (the reference's
    fp32 driver moves by ~4e-4 at 50 steps between BLAS thread partitions), so every fixture names its environment."""
    import hashlib, subprocess
    try:
        sha = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        sha = "unknown"
    par = torch.__config__.parallel_info()
    return dict(torch=torch.__version__, numpy=np.__version__, threads=torch.get_num_threads(),
                parallel_info_md5=hashlib.md5(par.encode()).hexdigest(), git=sha, cpu_count=os.cpu_count(),
                machine=" ".join(os.uname()[i] for i in (0, 2, 4)))


def save(name, **arrays):
    import json
    arrays = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()}
    arrays["provenance"] = np.array(json.dumps(provenance(), sort_keys=True))
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    print(f"  wrote {name}.npz  ({os.path.getsize(os.path.join(OUT, name + '.npz')) / 1024:.1f} KiB)")


def warped_mask_512(R, mask, coords):
    """U/editor.py:147-149 with text_embeddings fp16 (GPU path): coords rounded through fp16."""
    gt, wu = R.generic_torch, R.warp_utils
    image_mask = torch.from_numpy(mask[None]).tile((2, 1, 1))
    t = gt.reshape_transform_coords(torch.from_numpy(coords), in_mat_shape=image_mask.shape).tile(2, 1, 1, 1)
    t = t.half().float()
    out = wu.warp_grid_edit(image_mask[:, None], t, padding_mode="zeros", align_corners=True, mode="bilinear")
    return gt.binarize_tensor(out).float()          # [2,1,512,512]


def g_masks_and_warp(R):
    mask = cases.ellipse_mask()
    packs = {}
    for kind in ("translate", "rotate", "scale"):
        coords = cases.make_coords(kind, mask)
        m = warped_mask_512(R, mask, coords)
        assert torch.equal(m[0], m[1])
        packs[kind] = np.packbits(m[1, 0].numpy().astype(np.uint8))
        print(f"  warped mask {kind}: area {int(m[1].sum())} (input {int(mask.sum())})")
    save("G0_warped_mask_512", **packs)
    return packs


def unpack_mask(bits):
    m = np.unpackbits(bits)[: 512 * 512].reshape(512, 512).astype(np.float32)
    return torch.from_numpy(m)[None, None].tile(2, 1, 1, 1)


def g1_compute_attention(R):
    q, k, _ = cases.make_qkv(1, 1, 4, 64, 77, 16, spike=False)
    q, k = torch.from_numpy(q), torch.from_numpy(k)
    a = R.attention_sharing.compute_attention(q, k, 0.25)
    m1 = torch.zeros(1, 1, 8, 8); m1[..., 2:5, 2:5] = 1
    m2 = torch.zeros(1, 1, 8, 8); m2[..., 4:7, 1:4] = 1
    q2, k2, _ = cases.make_qkv(2, 1, 4, 64, 64, 16)
    q2, k2 = torch.from_numpy(q2), torch.from_numpy(k2)
    a_nomask = R.attention_sharing.compute_attention(q2, k2, 0.25)
    a_mask = R.attention_sharing.compute_attention(q2, k2, 0.25, None, fg_mask_warp=m1, fg_mask=m2, inpaint_mask=1 - m1)
    assert torch.equal(a_nomask, a_mask), "mask arguments are expected to be no-ops (SURVEY F2)"
    save("G1_compute_attention", attn=a, attn_self=a_nomask, mask_args_noop=np.array(1))


def g3_masks(R, packs):
    ap, gt = R.attention_processors, R.generic_torch
    mask = cases.ellipse_mask()
    out = {}
    for kind in ("translate", "rotate"):
        coords = torch.from_numpy(cases.make_coords(kind, mask))
        mnw = unpack_mask(packs[kind])
        amodal = gt.torch_erode(torch.from_numpy(cases.amodal_input(mask)))
        for S in (64, 32, 16, 8):
            f = 2
            q_base = torch.zeros(1, f, S * S, 4)
            q_img = torch.zeros(f, 4, S, S)
            image_mask = torch.from_numpy(mask[None]).tile((2, 1, 1))
            res = ap.process_and_cache_masks({}, S, image_mask, mnw, amodal, coords, q_base, q_img)
            names = ["mask_new_warped", "mask_warp", "amodal_mask", "mask_intersection", "mask_1_empty", "mask_wo_edit", "t_coords_q"]
            for n, t in zip(names, res[2:]):
                out[f"{kind}_{S}_{n}"] = t[:1] if n == "t_coords_q" else t
    save("G3_process_masks", **out)


def g5_interpolate(R):
    S, f, D = 32, 2, 8
    mask = cases.ellipse_mask()
    m = torch.from_numpy(mask)[None, None]
    m_s = (R.generic_torch.reshape_attention_mask(m, in_mat_shape=(1, S)) > 0.5) * 1.0
    fg = m_s[0, 0].reshape(-1)[None, None, :, None]
    feats = torch.from_numpy(cases.make_qkv(5, 1, f, S * S, S * S, D)[2])[None]
    dist = R.attention_processors.DISTANCE_CLASS.get_coord_distance(S, device="cpu")
    interp, w = R.attention_sharing.interpolate_from_mask(feats, fg, dist)
    sm = R.generic_torch.smooth_attention_features(feats)
    save("G5_interpolate", interp=interp, weights=w, smooth=sm, dist_row0=dist[0, 0], dist_row517=dist[0, 517])


def _make_controller(R, case, packs):
    ap, gt = R.attention_processors, R.generic_torch
    mask = cases.ellipse_mask()
    cls = ap.AttentionGeometryEdit if case["kind"] == "edit" else ap.AttentionGeometryRemover
    c = cls(["", ""], cases.NUM_STEPS, {"default_": 0.95}, cases.SELF_REPLACE, image_mask=mask,
            obj_edit_step=cases.OBJ_EDIT_STEP, device="cpu")
    c.amodal_mask = gt.torch_erode(torch.from_numpy(cases.amodal_input(mask)))
    c.mask_new_warped = unpack_mask(packs[case["coords"]])
    c.num_att_layers = 32
    c.cur_step = case["cur_step"]
    if case["cfg"]:
        c.coords_base, c.coords_edit, c.use_cfg = (2, 3), (3, 4), True
    else:
        c.coords_base, c.coords_edit, c.use_cfg = (0, 1), (1, 2), False
    return c


def g6_controller(R, packs):
    ap = R.attention_processors
    mask = cases.ellipse_mask()
    orig_rtc = ap.reshape_transform_coords
    for name, case in cases.CONTROLLER_CASES.items():
        print(" case", name)
        # ".type_as(q)" rounds the resampled coordinates through fp16 on the GPU path
        # (U/attention_processors.py:363); reproduce it on this fp32 CPU run when quant=True.
        if case["quant"]:
            ap.reshape_transform_coords = lambda *a, **k: orig_rtc(*a, **k).half().float()
        else:
            ap.reshape_transform_coords = orig_rtc
        c = _make_controller(R, case, packs)
        S, f, D = case["S"], case["f"], case["D"]
        N = S * S
        M = 77 if case["cross"] else N
        B = 4 if case["cfg"] else 2
        q, k, v = (torch.from_numpy(a) for a in cases.make_qkv(case["seed"], B, f, N, M, D))
        coords = torch.from_numpy(cases.make_coords(case["coords"], mask))
        grad_mode = not case["cfg"]
        if grad_mode:
            q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
        with torch.set_grad_enabled(grad_mode):
            out = c(q, k, v, is_cross=case["cross"], place_in_unet="up", transform_coords=coords, scale=D ** -0.5)
        rec = dict(out=out.detach(), cur_att_layer=np.array(c.cur_att_layer), cur_step=np.array(c.cur_step))
        if grad_mode:
            g_out = torch.from_numpy(np.random.default_rng(case["seed"] + 1000).standard_normal(tuple(out.shape), dtype=np.float32)) * 0.01
            total = (out * g_out).sum()
            if torch.is_tensor(c.loss):
                total = total + c.loss
                rec["loss"] = c.loss.detach()
                kind = "cross" if case["cross"] else "self"
                for key, val in c.loss_log_dict[kind].items():
                    rec["log_" + key] = val.detach() if torch.is_tensor(val) else np.array(val)
                rec["num_layers"] = np.array(c.loss_log_dict["num_layers"])
            dq, dk, dv = torch.autograd.grad(total, [q, k, v], allow_unused=True)
            rec.update(dq=dq, dk=dk if dk is not None else torch.zeros_like(k), dv=dv if dv is not None else torch.zeros_like(v))
        save("G6_" + name, **rec)
    ap.reshape_transform_coords = orig_rtc


def g7_counters(R, packs):
    case = dict(cases.CONTROLLER_CASES["edit_self_late_16"])
    c = _make_controller(R, case, packs)
    c.num_att_layers = 4
    c.cur_step = 46
    S, f, D = 8, 1, 8
    coords = torch.from_numpy(cases.make_coords("translate", cases.ellipse_mask()))
    trace = []
    with torch.no_grad():
        for call in range(14):
            q, k, v = (torch.from_numpy(a) for a in cases.make_qkv(70 + call, 4, f, S * S, S * S, D))
            c(q, k, v, is_cross=False, place_in_unet="mid", transform_coords=coords, scale=D ** -0.5)
            if call == 7:
                c.cur_step -= 1          # the driver's undo after an optimisation pass, U/editor.py:307
            trace.append((c.cur_att_layer, c.cur_step))
    save("G7_counters", trace=np.array(trace))


def g8_update_latent(R):
    rng = np.random.default_rng(8)
    lat = torch.from_numpy(rng.standard_normal((2, 4, 64, 64), dtype=np.float32))
    ctx = torch.from_numpy(rng.standard_normal((4, 77, 32), dtype=np.float32))
    wl = torch.from_numpy(rng.standard_normal((2, 4, 64, 64), dtype=np.float32))
    wc = torch.from_numpy(rng.standard_normal((4, 77, 32), dtype=np.float32))
    lat.requires_grad_(True); ctx.requires_grad_(True)
    loss = (lat * wl).sum() + (ctx * wc).pow(2).sum()
    mask = torch.from_numpy(cases.ellipse_mask())[None]          # controller.mask_new_warped[:1] has shape [1,1,H,W]; [:1] of [2,1,H,W]
    lat_o, ctx_o = R.optimization._update_latent(lat, loss, 0.37, mask[None][0], ctx)
    save("G8_update_latent", latents=lat_o.detach(), context=ctx_o.detach())


def g9_adaptive(R):
    opt = R.optimization
    import contextlib, io

    def run(fn, seq, weights):
        c = types.SimpleNamespace()
        c.default_loss_weights = {k: dict(v) for k, v in weights.items()}
        c.loss_weight_dict = c.default_loss_weights
        c.initialize_default_loss_weights = lambda c=c: setattr(c, "loss_weight_dict", c.default_loss_weights)
        traj = []
        for i, val in seq:
            with contextlib.redirect_stdout(io.StringIO()):
                fn(c, i, 2, {"self": {"removal": val}}, num_ddim_steps=50, removal_loss_value_in=-1.5)
            traj.append(c.loss_weight_dict["self"]["removal"])
        return np.array(traj)

    seq = [(0, -0.01), (2, -0.02), (4, -0.5), (6, -3.0), (8, -0.9), (10, -1.2), (12, -0.4), (14, -2.0), (16, -1.0),
           (18, -1.4), (20, -1.0), (22, -1.9), (24, -1.6), (26, -2.0), (28, -1.0), (30, -1.85), (32, -1.7), (40, -1.0), (44, 0.0)]
    w_e = {"self": {"sim": 55, "movement": 30.5, "removal": 2.6, "smoothness": 30, "amodal": 80.5},
           "cross": {"sim": 45, "movement": 30.34, "removal": 2.6, "smoothness": 15, "amodal": 3.5}}
    w_r = {"self": {"sim": 55, "removal": 4.6, "smoothness": 30}, "cross": {"sim": 45, "removal": 4.6, "smoothness": 15}}
    save("G9_adaptive", seq=np.array(seq), edit=run(opt.adaptive_optimization_step_editing, seq, w_e),
         remover=run(opt.adaptive_optimization_step_remover, seq, w_r))


def g10_ddim(R):
    import GeoDiffuser.utils.inversion as inv
    sys.path.insert(0, HERE)
    import ref_cpu
    ac = ref_cpu.alphas_cumprod()
    fake = types.SimpleNamespace()
    fake.scheduler = types.SimpleNamespace(config=types.SimpleNamespace(num_train_timesteps=1000), num_inference_steps=50,
                                           alphas_cumprod=ac, final_alpha_cumprod=ac[0])
    rng = np.random.default_rng(10)
    x = torch.from_numpy(rng.standard_normal((1, 4, 8, 8), dtype=np.float32))
    e = torch.from_numpy(rng.standard_normal((1, 4, 8, 8), dtype=np.float32))
    prev = torch.stack([inv.NullInversion.prev_step(fake, e, int(t), x) for t in range(980, -1, -20)])
    nxt = torch.stack([inv.NullInversion.next_step(fake, e, int(t), x) for t in range(0, 1000, 20)])
    save("G10_ddim", prev=prev, next=nxt, alphas_cumprod=ac)


def g11_geometry(R):
    import GeoDiffuser.utils.vis_utils as vis
    size = 64
    mask = cases.ellipse_mask(cx=30, cy=33, ax=11, ay=9, size=size)
    v, u = np.mgrid[0:size, 0:size].astype(np.float32)
    depth = np.where(mask > 0.5, 0.5 + 0.2 * (u / size - 0.5), 0.9).astype(np.float32)
    image = np.zeros((size, size, 3), dtype=np.float32)
    out = {}
    tf = {"translate": vis.translateMatrix(0.1, -0.05, 0.02),
          "rotate": vis.rotateAxis(25.0, 1).float(),
          "mixed": (vis.translateMatrix(0.05, 0.0, 0.05) @ vis.rotateAxis(-15.0, 1).float() @ vis.rotateAxis(10.0, 2).float())}
    for name, T in tf.items():
        t_coords, _ = vis.get_transform_coordinates(image, depth.copy(), mask, transform_in=T.float(), focal_length=550 * size / 512.0, return_mesh=False)
        out[name] = t_coords
        out[name + "_T"] = T.numpy()
    const = np.ones((size, size), dtype=np.float32) * 0.5
    t_coords, _ = vis.get_transform_coordinates(image, const, mask, transform_in=vis.translateMatrix(0.1, 0, 0), focal_length=550 * size / 512.0, return_mesh=False)
    out["const_depth"] = t_coords
    save("G11_geometry", **out)


def g12_mesh(R):
    """get_mesh / create_triangles / get_coordinate_array / get_indexing_grid (U/warp_utils.py:304-399) — the reference's own
    code (pure torch) on masks that tell the per-TRIANGLE corner test from a per-quad one (holes, one-pixel notches, thin bars, a
    diagonal staircase), with the projected coordinates of G11's transforms: vertices and face lists as the reference hands them to
    pytorch3d's ``Meshes`` (captured by a recording stand-in)."""
    import GeoDiffuser.utils.vis_utils as vis
    wu = R.warp_utils
    captured = {}

    class Rec:
        def __init__(self, verts=None, faces=None, textures=None):
            captured["verts"], captured["faces"] = verts[0].detach().clone(), faces[0].detach().clone()

    old_meshes, old_tex = wu.Meshes, wu.TexturesVertex
    wu.Meshes, wu.TexturesVertex = Rec, (lambda **k: None)
    out = {}
    try:
        size = 64
        for name, mask in cases.mesh_masks(size).items():
            v, u = np.mgrid[0:size, 0:size].astype(np.float32)
            depth = np.where(mask > 0.5, 0.5 + 0.2 * (u / size - 0.5), 0.9).astype(np.float32)
            T = (vis.translateMatrix(0.05, 0.0, 0.05) @ vis.rotateAxis(-15.0, 1).float() @ vis.rotateAxis(10.0, 2).float()).float()
            image = np.zeros((size, size, 3), dtype=np.float32)
            t_coords, _ = vis.get_transform_coordinates(image, depth.copy(), mask, transform_in=T, focal_length=550 * size / 512.0,
                                                        return_mesh=False)
            tc = torch.from_numpy(t_coords)[None].permute(0, 3, 1, 2)                 # b, 3, h, w as at U/warp_utils.py:456
            wu.get_mesh(tc, torch.from_numpy(mask)[None, None].float())
            out[name + "_verts"] = captured["verts"].numpy()
            out[name + "_faces"] = captured["faces"].numpy().astype(np.int32)
            out[name + "_t_coords"] = t_coords
    finally:
        wu.Meshes, wu.TexturesVertex = old_meshes, old_tex
    save("G12_mesh", **out)


def g13_resample(R):
    gt = R.generic_torch
    mask = cases.ellipse_mask()
    coords = torch.from_numpy(cases.make_coords("rotate", mask))
    out = {}
    for S in (64, 8):
        out[f"coords_{S}"] = gt.reshape_transform_coords(coords, in_mat_shape=(1, 1, S, S))
        out[f"mask_{S}"] = gt.reshape_attention_mask(torch.from_numpy(mask)[None, None], in_mat_shape=(1, S))
    blk = torch.zeros(1, 1, 9, 9); blk[..., 3:6, 3:6] = 1
    out["erode3"] = gt.torch_erode(blk, 3); out["dilate3"] = gt.torch_dilate(blk, 3); out["dilate5"] = gt.torch_dilate(blk, 5)
    out["gauss_w"] = gt.GAUSSIAN_FEATURE_SMOOTHER.weight[0, 0]
    save("G13_resample", **out)


def g4_losses(R, packs):
    """Loss values and d/d(replace_out), d/d(replace_att) on seeded features (S=32, f=2, D=8)."""
    ap, ls = R.attention_processors, R.loss
    S, f, D = 32, 2, 8
    N = S * S
    rng = np.random.default_rng(4)
    eo = torch.from_numpy(rng.standard_normal((1, f, N, D), dtype=np.float32))
    ro = torch.from_numpy(rng.standard_normal((1, f, N, D), dtype=np.float32)).requires_grad_(True)
    m_edit = torch.from_numpy((rng.random((1, 1, N, 1)) > 0.8).astype(np.float32) * rng.choice([0.25, 0.5, 1.0], size=(1, 1, N, 1)).astype(np.float32))
    m_wo = torch.from_numpy((rng.random((1, 1, N, 1)) > 0.4).astype(np.float32))
    m_inp = torch.zeros(1, 1, N, 1); m_inp[0, 0, 300:340] = 1
    m_amo = torch.from_numpy((rng.random((1, 1, N, 1)) > 0.9).astype(np.float32))
    dist = ap.DISTANCE_CLASS.get_coord_distance(S, device="cpu")
    a_e = torch.softmax(torch.from_numpy(rng.standard_normal((f, N, N), dtype=np.float32)) * 2, -1).requires_grad_(True)
    a_b = torch.softmax(torch.from_numpy(rng.standard_normal((f, N, N), dtype=np.float32)) * 2, -1)
    l_bg = ap.background_preservation_loss(eo, ro, m_wo)
    l_mv = ap.object_placement_loss_geodiff(eo, ro, m_edit)
    l_am = ap.amodal_loss_geodiff(eo, ro, m_edit, dist, m_amo)
    l_sm, _, _ = ls.get_smoothness_loss(ro)
    l_rm = ap.removal_loss_geodiff(a_e, a_b, m_inp, m_wo, dist, f)
    g = {}
    for nm, l in (("bg", l_bg), ("mv", l_mv), ("am", l_am), ("sm", l_sm)):
        g["d_" + nm] = torch.autograd.grad(l, ro, retain_graph=True)[0]
    d_rm = torch.autograd.grad(l_rm, a_e)[0]
    save("G4_losses", bg=l_bg.detach(), mv=l_mv.detach(), am=l_am.detach(), sm=l_sm.detach(), rm=l_rm.detach(),
         d_rm_rows=d_rm[:, 300:340], **g)


def g14_histogram(R):
    """masked_histogram_matching on the seeded images / masks of cases.hist_case (float64 out, exact)."""
    ip = R.image_processing
    out = {}
    for name in cases.HIST_CASES:
        src, tmpl, m, ms = cases.hist_case(name)
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):          # the reference prints "iden mask" for mask=None
            out[name] = ip.masked_histogram_matching(src, tmpl, m, ms)
    save("G14_histogram", **out)


def g15_exp_folder():
    """Experiment folders written by the reference's ``save_exp`` (committed as data under tests/golden/exp_root/), what its
    ``read_exp`` returns for them, and the 4x4 transforms ``get_transformed_mask`` composes (``project_image`` intercepted)."""
    import shutil
    U = ref_import.import_reference_ui()
    root = os.path.join(OUT, "exp_root")
    shutil.rmtree(root, ignore_errors=True)
    out = {}
    for cat, idx in cases.EXP_CASES:
        e = cases.exp_case(cat, idx)
        U.save_exp(root, e["image"], e["depth"], e["depth_vis"], e["mask"], e["transform"], transformed_image=e.get("transformed"),
                   background_image=e.get("background"), h=e["h"], w=e["w"], exp_transform_type=cat)
        d = U.read_exp(os.path.join(root, cat, str(idx)))
        for k, v in d.items():
            if isinstance(v, np.ndarray):
                out[f"{cat}_{idx}__{k}"] = v
        out[f"{cat}_{idx}__none_keys"] = np.array(sorted(k for k, v in d.items() if v is None))
    out["is_root"] = np.array(U.check_if_exp_root(root))
    out["is_root_leaf"] = np.array(U.check_if_exp_root(os.path.join(root, "Mix")))
    captured = {}
    U.project_image = lambda *a, **k: captured.setdefault("t", a[4])
    for i, kw in enumerate(cases.TRANSFORM_CASES):
        captured.clear()
        U.get_transformed_mask(None, None, None, None, kw.get("translation_x", 0.0), kw.get("translation_y", 0.0),
                               kw.get("translation_z", 0.0), kw.get("rotation_x", 0.0), kw.get("rotation_y", 0.0),
                               kw.get("rotation_z", 0.0), None, 1.3, scale_x=kw.get("scale_x", 1.0), scale_y=kw.get("scale_y", 1.0),
                               scale_z=kw.get("scale_z", 1.0))
        out[f"transform_{i}"] = captured["t"].numpy()
    save("G15_exp_folder", **out)


def g17_attention_store(R, packs):
    """store_attention_maps: what the reference's AttentionGeometryEdit leaves in attention_store after two 3-layer "steps"
    (self 16^2 down, cross 16^2 mid, self 32^2 up — the last one is above the 16^2 cut and must not be stored)."""
    case = dict(cases.CONTROLLER_CASES["edit_self_cfg_32"])
    c = _make_controller(R, case, packs)
    c.num_att_layers, c.cur_step, c.store_attention_maps = 3, 0, True
    coords = torch.from_numpy(cases.make_coords(case["coords"], cases.ellipse_mask()))
    out = {}
    with torch.no_grad():
        for step in range(2):
            for li, (S, cross, place) in enumerate(cases.STORE_LAYERS):
                N = S * S
                q, k, v = (torch.from_numpy(a) for a in cases.make_qkv(900 + 10 * step + li, 4, 2, N, 77 if cross else N, 16))
                c(q, k, v, is_cross=cross, place_in_unet=place, transform_coords=coords, scale=0.25)
    for key, val in c.attention_store.items():
        if isinstance(val, list):
            out["n__" + key] = np.array(len(val))
            for i, a in enumerate(val):
                out[f"{key}__{i}"] = a
        else:
            out["int__" + key] = np.array(val)
    avg = c.get_average_attention() if all(isinstance(v, list) for v in c.attention_store.values()) else None
    out["cur_step"] = np.array(c.cur_step)
    save("G17_attention_store", **out)


def run_reference_loop(R, kind="geometry_editor", cfg=None, prepare=None, x_T_eps=0.0, tiny=True, sd14=False, sdxl=False):
    """The reference's own per-step driver (``text2image_ldm_stable``, U/editor.py:65-423: optimisation pass -> _update_latent ->
    adaptive schedule -> CFG pass -> reference-latent replacement -> latent warp) with its own processors / controller, driving the
    narrow SD-topology UNet of geodiffuser_amd (seeded random weights, fp32, CPU) through a CPU DDIM scheduler built from the
    reference's closed form.  -> (final latents, loss log, controller, pipe).
    ``prepare(pipe)``: optional hook applied to the freshly built model (oracle/fp16_emulation.py installs 16-bit rounding there);
    ``x_T_eps``: relative perturbation of the start latent (sensitivity probe)."""
    import GeoDiffuser.utils.editor as RE
    from types import SimpleNamespace
    from geodiffuser_amd.pipeline import build_random_sd21
    import ref_cpu as O
    c = cfg or cases.LOOP
    inp = cases.loop_inputs(c)
    # tiny=False: the full SD2.1-base width (865 M parameters); sd14: the SD1.x head layout (head dims 40 / 80 / 160) of the reference's default model
    if sdxl:      # SDXL-base topology (narrow): its text_time conditioning comes from unet.default_added_cond, the reference's call signature is unchanged
        from geodiffuser_amd.pipeline import build_random_sdxl
        pipe = build_random_sdxl(device="cpu", dtype=torch.float32, tiny=True, image_size=c["size"])
    else:
        pipe = build_random_sd21(device="cpu", dtype=torch.float32, tiny=tiny, sd14=sd14)
    if prepare is not None:
        prepare(pipe)

    class CpuDDIM:
        def __init__(self):
            self.alphas_cumprod = O.alphas_cumprod()

        def set_timesteps(self, n):
            self.num_inference_steps = n
            self.timesteps = torch.from_numpy(O.ddim_timesteps(n))

        def step(self, eps, t, x, eta=0.0):
            return {"prev_sample": O.prev_step(eps, int(t), x, self.alphas_cumprod, self.num_inference_steps)}

    model = SimpleNamespace(unet=pipe.unet, vae=pipe.vae, tokenizer=pipe.tokenizer, text_encoder=pipe.text_encoder, scheduler=CpuDDIM(),
                            device=torch.device("cpu"))
    ap = R.attention_processors
    ap.USE_PEFT_BACKEND = True                 # plain nn.Linear projections: no LoRA scale argument (diffusers' PEFT branch)
    orig_rtc = ap.reshape_transform_coords
    RE.IMAGE_SIZE, RE.NUM_DDIM_STEPS, RE.GUIDANCE_SCALE, RE.SKIP_OPTIM_STEPS, RE.PROGRESS_BAR = c["size"], c["steps"], c["guidance"], c["skip_optim"], None
    mask = torch.from_numpy(inp["mask"])
    coords = torch.from_numpy(inp["coords"])
    if kind == "geometry_editor":
        lw = {"self": {"sim": 55, "movement": 30.5, "removal": 2.6, "smoothness": 30.0, "amodal": 80.5},
              "cross": {"sim": 45, "movement": 30.34, "removal": 2.6, "smoothness": 15.0, "amodal": 3.5}}
        ctrl = ap.AttentionGeometryEdit(["", ""], c["steps"], {"default_": c["cross_replace"]}, c["self_replace"], image_mask=inp["mask"],
                                        obj_edit_step=c["obj_edit_step"], device="cpu")
        ctrl.amodal_mask = R.generic_torch.torch_erode(torch.from_numpy(cases.amodal_input(inp["mask"], *c.get("amodal_shift", (32, -12)))))
    else:
        lw = {"self": {"sim": 55, "removal": 4.6, "smoothness": 30.0}, "cross": {"sim": 45, "removal": 4.6, "smoothness": 15.0}}
        ctrl = ap.AttentionGeometryRemover(["", ""], c["steps"], {"default_": 0.9}, 0.9, image_mask=inp["mask"], obj_edit_step=1.0,
                                           device="cpu")
    ctrl.default_loss_weights = lw
    ctrl.initialize_default_loss_weights()
    ddim = [torch.from_numpy(a) for a in inp["ddim_latents"]]
    x_T = torch.from_numpy(inp["x_T"])
    if x_T_eps:
        g = torch.Generator().manual_seed(5)
        x_T = x_T * (1.0 + x_T_eps * torch.randn(x_T.shape, generator=g))
    # record the latent update of every optimisation pass (-step * masked gradient): isolates ONE backward pass from the loop
    updates = []
    weights = []           # adaptive removal weight in effect at each optimisation pass (U/optimization.py:7-105 edits it after the pass)
    orig_update = RE._update_latent

    def rec_update(latents, loss, step_size, mask=None, context=None, **kw):
        res = orig_update(latents, loss, step_size, mask, context, **kw)
        updates.append((res[0][-1:].detach() - latents[-1:].detach()).clone())
        weights.append(float(ctrl.loss_weight_dict["self"]["removal"]))
        return res

    RE._update_latent = rec_update
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        lat, _, log = RE.text2image_ldm_stable(model, ["", ""], ctrl, latent=x_T, num_inference_steps=c["steps"],
                                               guidance_scale=c["guidance"], uncond_embeddings=None, transform_coordinates=coords,
                                               mask_obj=mask, optimize_steps=c["optimize_steps"], latent_replace=c["latent_replace"], lr=c["lr"],
                                               optimize_embeddings=True, optimize_latents=True, ddim_latents=ddim, ddim_noise=None,
                                               edit_type=kind, fast_start_steps=0.0, num_first_optim_steps=1,
                                               use_adaptive_optimization=True, return_type="latents")
    ap.reshape_transform_coords = orig_rtc
    RE._update_latent = orig_update
    ctrl._recorded_updates = updates
    ctrl._recorded_weights = weights
    return lat.detach(), log, ctrl, pipe


def g18_loop(R, kind="geometry_editor", cfg=None, name=None, tiny=True, sd14=False, sdxl=False, record_weights=False):
    """Records the final latents and the loss log of every optimisation step of run_reference_loop."""
    lat, log, ctrl, pipe = run_reference_loop(R, kind, cfg, tiny=tiny, sd14=sd14, sdxl=sdxl)
    out = {"latents": lat.detach(), "steps": np.array(sorted(log))}
    for i, d in log.items():
        for att in ("self", "cross"):
            for k, v in d[att].items():
                out[f"log_{i}_{att}_{k}"] = np.array(float(v))
        out[f"log_{i}_num_layers"] = np.array(d["num_layers"])
    w = torch.cat([p.detach().reshape(-1)[:64] for p in pipe.unet.parameters()])
    out["weight_probe"] = w                                         # to recognise the same seeded weights on the test machine
    out["final_weights_self_removal"] = np.array(float(ctrl.loss_weight_dict["self"]["removal"]))
    out["first_update"] = ctrl._recorded_updates[0].numpy()          # latent update of the first optimisation pass
    if record_weights:                                               # (G28 / G29: the whole trajectory of the adaptive schedule)
        out["weights_self_removal"] = np.array(ctrl._recorded_weights, dtype=np.float64)
    save(name or ("G18_loop" if kind == "geometry_editor" else "G19_loop_remover"), **out)


def g25_null_text(R):
    """Null-text optimisation (U/inversion.py:213-259): the reference's own ``NullInversion.null_optimization`` on CPU (fp32) over the
    narrow UNet (plain torch attention processor), a seeded 4-step trajectory, 3 inner Adam steps per DDIM step."""
    from types import SimpleNamespace
    import GeoDiffuser.utils.inversion as RI
    from geodiffuser_amd.pipeline import build_random_sd21
    import ref_cpu as O
    from fp16_emulation import _CpuVanillaProcessor
    pipe = build_random_sd21(device="cpu", dtype=torch.float32, tiny=True)
    pipe.unet.set_attn_processor(_CpuVanillaProcessor())
    c = cases.NULL_TEXT
    traj = cases.null_text_inputs()

    class CpuDDIM:
        def __init__(self, n):
            self.alphas_cumprod = O.alphas_cumprod()
            self.final_alpha_cumprod = self.alphas_cumprod[0]
            self.config = SimpleNamespace(num_train_timesteps=1000)
            self.num_inference_steps = n
            self.timesteps = torch.from_numpy(O.ddim_timesteps(n))

    model = SimpleNamespace(unet=pipe.unet, scheduler=CpuDDIM(c["steps"]), tokenizer=pipe.tokenizer, text_encoder=pipe.text_encoder,
                            device=torch.device("cpu"))
    import tqdm as _tqdm
    RI.tqdm = _tqdm.std.tqdm                 # the reference imports the notebook progress bar (needs ipywidgets)
    # The reference never imports the optimiser it names (U/inversion.py:223 raises NameError: its own default
    # perform_inversion=True cannot run; every driver passes False).  The prompt-to-prompt code this method was taken from uses
    # torch.optim.adam.Adam and torch.nn.functional as nnf: supplied here so that the method's arithmetic can be recorded.
    if not hasattr(RI, "Adam"):
        RI.Adam = torch.optim.Adam
    if not hasattr(RI, "nnf"):
        RI.nnf = torch.nn.functional
    ni = object.__new__(RI.NullInversion)
    ni.model, ni.num_ddim_steps, ni.guidance_scale = model, c["steps"], c["guidance"]
    tok = pipe.tokenizer
    ids = tok([""], padding="max_length", max_length=tok.model_max_length, return_tensors="pt").input_ids
    with torch.no_grad():
        e = pipe.text_encoder(ids)[0]
    ni.context = torch.cat([e, e])
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        out = ni.null_optimization([torch.from_numpy(a) for a in traj], c["inner"], c["eps"])
    save("G25_null_text", uncond=torch.cat(out).detach().numpy(), context0=e.detach().numpy(),
         weight_probe=torch.cat([p.detach().reshape(-1)[:64] for p in pipe.unet.parameters()]).numpy())


def g16_batch_config():
    """What ``perform_exp`` hands to ``perform_geometric_edit`` for each live edit type (call intercepted) -> JSON."""
    import json
    L = ref_import.import_reference_batch_driver()
    got = {}
    e = cases.exp_case("Mix", 1)
    exp_dict = {"input_image_png": e["image"], "input_mask_png": np.repeat((e["mask"] * 255).astype(np.uint8)[..., None], 3, -1),
                "depth_npy": e["depth"], "transform_npy": e["transform"]}
    for etype in ("geometry_editor", "geometry_remover"):
        seen = {}

        def fake(image, depth, image_mask, transform_in, prompt, **kw):
            seen.update(kw)
            seen["_mask_sum"] = float(image_mask.sum())
            seen["_mask_shape"] = list(image_mask.shape)
            seen["_transform_dtype"] = str(transform_in.dtype)
            return [image, image], {}

        L.perform_geometric_edit = fake
        L.perform_exp(exp_dict, edit_type=etype)
        got[etype] = {k: v for k, v in seen.items() if k not in ("ldm_stable_model", "tokenizer_model", "scheduler_in")}
    with open(os.path.join(OUT, "G16_batch_config.json"), "w") as fh:
        json.dump(got, fh, indent=1, sort_keys=True)
    print("  wrote G16_batch_config.json")


# name -> (kind, cases.<cfg>, g18_loop kwargs, approx CPU seconds on 8 cores).  Every loop fixture is the reference's own driver
# (run_reference_loop); the table is what `--check` and the CPU suite iterate over.
LOOP_FIXTURES = {
    "G18_loop": ("geometry_editor", "LOOP", {}, 8),
    "G19_loop_remover": ("geometry_remover", "LOOP", {}, 8),
    "G20_loop_cfg0": ("geometry_editor", "LOOP_CFG0", {}, 20),
    # BASELINE configs[0] (256^2, 2-D translation, 20-step DDIM) at the FULL SD2.1-base width (865 M-parameter random-init UNet, fp32)
    "G21_loop_cfg0_full": ("geometry_editor", "LOOP_CFG0", dict(tiny=False), 400),
    # BASELINE configs[1] SHAPE (512^2, 3-D rotation), full width, 4 DDIM steps (the reference materialises [10, 4096, 4096] fp32 maps)
    "G22_loop_cfg1_full": ("geometry_editor", "LOOP_CFG1", dict(tiny=False), 300),
    # SD1.x head layout (head dims 40 / 80 / 160: the reference's default model, U/editor.py:58), narrow
    "G23_loop_sd14": ("geometry_editor", "LOOP", dict(sd14=True), 10),
    # the removal edit at the full SD2.1-base width (256^2, 6 steps)
    "G26_loop_remover_full": ("geometry_remover", "LOOP", dict(tiny=False), 200),
    # SDXL-topology UNet (narrow) at 512^2
    "G27_loop_sdxl": ("geometry_editor", "LOOP_SDXL", dict(sdxl=True), 60),
    # BASELINE configs[1] at its stated length (50 steps, 17 optimisation passes), narrow model
    "G28_loop_cfg1_t50": ("geometry_editor", "LOOP_CFG1_T50", dict(record_weights=True), 120),
    # BASELINE configs[3] at its stated length (768^2 removal, 75-step schedule), narrow model, eps-prediction
    "G29_loop_remover768_t75": ("geometry_remover", "LOOP_REM768_T75", dict(record_weights=True), 300),
    # BASELINE configs[1] ITSELF: full width x full length (865 M parameters, 512^2, 3-D rotation, 50 steps, 17 optimisation passes, the
    # batch driver's geometry_editor column large_scale_editor.py:264-299) -- the workload bench.py's headline is quoted on
    "G30_loop_cfg1_full_t50": ("geometry_editor", "LOOP_CFG1_T50", dict(tiny=False, record_weights=True), 6000),
}
GEN_THREADS = 8            # pinned: the fp32 loop fixtures depend on the BLAS thread partition


def _ref_for_loops():
    R = ref_import.import_reference()
    sys.path.insert(0, ROOT)
    torch.set_num_threads(GEN_THREADS)
    return R


def gen_loop(R, name):
    kind, cfg, kw, _ = LOOP_FIXTURES[name]
    g18_loop(R, kind, getattr(cases, cfg), name, **kw)


def check(names):
    """Regenerate fixtures into a scratch directory and compare with the committed files: bit-identical -> 0; otherwise print the
    largest relative distance per array and return 1 (used by tests/test_oracle_golden.py for the fixtures that take < 60 s)."""
    global OUT
    import tempfile
    committed = OUT
    bad = 0
    R = _ref_for_loops()
    with tempfile.TemporaryDirectory() as tmp:
        OUT = tmp
        for name in names:
            gen_loop(R, name)
            a, b = np.load(os.path.join(committed, name + ".npz")), np.load(os.path.join(tmp, name + ".npz"))
            keys = sorted((set(a.files) | set(b.files)) - {"provenance"})
            worst = ("", 0.0)
            for k in keys:
                if k not in a.files or k not in b.files:
                    worst = (k + " (missing)", float("inf")); break
                x, y = np.asarray(a[k], dtype=np.float64), np.asarray(b[k], dtype=np.float64)
                if x.shape != y.shape:
                    worst = (k + " (shape)", float("inf")); break
                d = float(np.linalg.norm(x - y) / (np.linalg.norm(x) + 1e-30))
                if d > worst[1]:
                    worst = (k, d)
            prov = str(a["provenance"]) if "provenance" in a.files else "(no provenance recorded)"
            print(f"CHECK {name}: " + ("bit-identical" if worst[1] == 0.0 else f"DIFFERS, worst {worst[0]} rel {worst[1]:.3e}") +
                  f"   committed from {prov}")
            bad |= worst[1] != 0.0
    OUT = committed
    return int(bad)


def main():
    args = sys.argv[1:]
    if args and args[0] == "--check":
        sys.exit(check(args[1:] or [n for n, v in LOOP_FIXTURES.items() if v[3] < 60]))
    if args and args[0] == "--loops":                     # every loop fixture except the hour-long G30
        R = _ref_for_loops()
        for n in LOOP_FIXTURES:
            if n != "G30_loop_cfg1_full_t50":
                print(n); gen_loop(R, n)
        return
    short = {n.split("_")[0]: n for n in LOOP_FIXTURES}
    if args and (args[0] in LOOP_FIXTURES or args[0] in short):
        R = _ref_for_loops()
        names = [short.get(a, a) for a in args]
        if args == ["G18"]:                               # historical shorthand: the three quick loops
            names = ["G18_loop", "G19_loop_remover", "G20_loop_cfg0"]
        for n in names:
            print(n); gen_loop(R, n)
        return
    if args and args[0] == "G15":
        os.makedirs(OUT, exist_ok=True)
        print("G15"); g15_exp_folder()
        print("G16"); g16_batch_config()
        return
    if args and args[0] == "G17":
        R = ref_import.import_reference()
        print("G17"); g17_attention_store(R, g_masks_and_warp(R))
        return
    if args and args[0] == "G25":
        R = ref_import.import_reference()
        sys.path.insert(0, ROOT)
        print("G25"); g25_null_text(R)
        return
    if args and args[0] == "G12":
        R = ref_import.import_reference()
        print("G12"); g12_mesh(R)
        return
    if args and args[0] == "G14":
        R = ref_import.import_reference()
        os.makedirs(OUT, exist_ok=True)
        print("G14"); g14_histogram(R)
        return
    torch.manual_seed(0)
    torch.set_num_threads(8)
    R = ref_import.import_reference()
    os.makedirs(OUT, exist_ok=True)
    print("G0 warped masks"); packs = g_masks_and_warp(R)
    print("G1"); g1_compute_attention(R)
    print("G3"); g3_masks(R, packs)
    print("G4"); g4_losses(R, packs)
    print("G5"); g5_interpolate(R)
    print("G6"); g6_controller(R, packs)
    print("G7"); g7_counters(R, packs)
    print("G8"); g8_update_latent(R)
    print("G9"); g9_adaptive(R)
    print("G10"); g10_ddim(R)
    print("G11"); g11_geometry(R)
    print("G12"); g12_mesh(R)
    print("G13"); g13_resample(R)
    print("G14"); g14_histogram(R)
    print("G17"); g17_attention_store(R, packs)
    print("G15"); g15_exp_folder()


if __name__ == "__main__":
    main()

"""Pins the CPU oracle (oracle/ref_cpu.py) against golden vectors recorded from the reference's own
Python (oracle/gen_golden.py).  CPU only.  fp32-vs-fp32: tolerance 1e-5 relative (max-norm), exact
for integers / masks / counters."""
import os

import numpy as np
import pytest
import torch

import cases
import ref_cpu as O
from _util import case_gout, case_inputs, fixture_mismatch, load, rel_err, warped_mask

TOL = 1e-5


def test_g1_compute_attention():
    g = load("G1_compute_attention")
    q, k, _ = cases.make_qkv(1, 1, 4, 64, 77, 16, spike=False)
    a = O.compute_attention(torch.from_numpy(q), torch.from_numpy(k), 0.25)
    assert rel_err(a, g["attn"]) < TOL
    q2, k2, _ = cases.make_qkv(2, 1, 4, 64, 64, 16)
    a2 = O.compute_attention(torch.from_numpy(q2), torch.from_numpy(k2), 0.25)
    assert rel_err(a2, g["attn_self"]) < TOL
    assert int(g["mask_args_noop"]) == 1          # SURVEY F2 recorded at generation time


@pytest.mark.parametrize("kind", ["translate", "rotate"])
def test_g3_process_masks(kind):
    g = load("G3_process_masks")
    mask = cases.ellipse_mask()
    coords = torch.from_numpy(cases.make_coords(kind, mask))
    amodal = O.torch_erode(torch.from_numpy(cases.amodal_input(mask)))
    image_mask = torch.from_numpy(mask[None]).tile((2, 1, 1))
    for S in (64, 32, 16, 8):
        r = O.process_masks(image_mask, warped_mask(kind), amodal, coords, S, 2, coords_quant=None)
        for n in ("mask_new_warped", "mask_warp", "amodal_mask", "mask_intersection", "mask_1_empty", "mask_wo_edit"):
            assert np.array_equal(r[n].numpy(), g[f"{kind}_{S}_{n}"]), (S, n)
        assert np.array_equal(r["t_coords_q"][:1].numpy(), g[f"{kind}_{S}_t_coords_q"]), S
        # resampled binary masks only take values in {0, 1/4, 1/2, 3/4, 1}
        assert set(np.unique(r["mask_new_warped"].numpy())) <= {0.0, 0.25, 0.5, 0.75, 1.0}


def test_g4_losses():
    g = load("G4_losses")
    S, f, D = 32, 2, 8
    N = S * S
    rng = np.random.default_rng(4)
    eo = torch.from_numpy(rng.standard_normal((1, f, N, D), dtype=np.float32))
    ro = torch.from_numpy(rng.standard_normal((1, f, N, D), dtype=np.float32)).requires_grad_(True)
    m_edit = torch.from_numpy((rng.random((1, 1, N, 1)) > 0.8).astype(np.float32) * rng.choice([0.25, 0.5, 1.0], size=(1, 1, N, 1)).astype(np.float32))
    m_wo = torch.from_numpy((rng.random((1, 1, N, 1)) > 0.4).astype(np.float32))
    m_inp = torch.zeros(1, 1, N, 1); m_inp[0, 0, 300:340] = 1
    m_amo = torch.from_numpy((rng.random((1, 1, N, 1)) > 0.9).astype(np.float32))
    dist = O.coord_distance(S)
    a_e = torch.softmax(torch.from_numpy(rng.standard_normal((f, N, N), dtype=np.float32)) * 2, -1).requires_grad_(True)
    a_b = torch.softmax(torch.from_numpy(rng.standard_normal((f, N, N), dtype=np.float32)) * 2, -1)
    vals = dict(bg=O.background_preservation_loss(eo, ro, m_wo), mv=O.object_placement_loss(eo, ro, m_edit),
                am=O.amodal_loss(eo, ro, m_edit, dist, m_amo), sm=O.smoothness_loss(ro),
                rm=O.removal_loss(a_e, a_b, m_inp, m_wo, dist, f))
    for k_, v_ in vals.items():
        assert abs(float(v_) - float(g[k_])) <= TOL * max(1.0, abs(float(g[k_]))), k_
    for k_ in ("bg", "mv", "am", "sm"):
        d = torch.autograd.grad(vals[k_], ro, retain_graph=True)[0]
        assert rel_err(d, g["d_" + k_]) < TOL, k_
    d_rm = torch.autograd.grad(vals["rm"], a_e)[0]
    assert rel_err(d_rm[:, 300:340], g["d_rm_rows"]) < TOL
    assert float(d_rm[:, :300].abs().max()) == 0.0


def test_g5_interpolate_and_smooth():
    g = load("G5_interpolate")
    S, f, D = 32, 2, 8
    mask = cases.ellipse_mask()
    m_s = (O.reshape_attention_mask(torch.from_numpy(mask)[None, None], S) > 0.5) * 1.0
    fg = m_s[0, 0].reshape(-1)[None, None, :, None]
    feats = torch.from_numpy(cases.make_qkv(5, 1, f, S * S, S * S, D)[2])[None]
    dist = O.coord_distance(S)
    interp, w = O.interpolate_from_mask(feats, fg, dist)
    assert rel_err(interp, g["interp"]) < TOL
    assert rel_err(w, g["weights"]) < TOL
    assert rel_err(O.smooth_attention_features(feats), g["smooth"]) < TOL
    assert np.array_equal(dist[0, 0].numpy(), g["dist_row0"])
    assert np.array_equal(dist[0, 517].numpy(), g["dist_row517"])


def test_deterministic_nearest4_rule_keeps_the_reference_distances():
    """ref_cpu.nearest4_by_index (test aid: ascending (background, exact integer squared distance, index)) picks, for every pixel, four
    pixels at exactly the inverse distances the reference's torch.topk keeps (pinned by G5 above) — it only resolves WHICH of several
    equidistant pixels is taken — and the distance-decay weights are identical."""
    for S in (32, 64):
        mask = cases.ellipse_mask()
        m_s = (O.reshape_attention_mask(torch.from_numpy(mask)[None, None], S) > 0.5) * 1.0
        fg = m_s[0, 0].reshape(-1)[None, None, :, None]
        dist = O.coord_distance(S)
        d_new = dist * 512 / 2.0 + 100000 * (1.0 - (fg[:1, :1, :, 0] > 0.5) * 1.0)
        inv = 1.0 / (d_new + 1e-4)
        top = torch.topk(inv, k=4, dim=-1, largest=True, sorted=False).values[0].sort(-1).values
        idx = O.nearest4_by_index(fg[0, 0, :, 0], S)
        got = torch.gather(inv[0], 1, idx).sort(-1).values
        assert rel_err(got, top) < 1e-6
        assert bool((fg[0, 0, :, 0][idx] > 0.5).all())                       # the ellipse has >= 4 foreground pixels: never a background pick
        feats = torch.from_numpy(cases.make_qkv(5, 1, 2, S * S, S * S, 8)[2])[None]
        _, w_a = O.interpolate_from_mask(feats, fg, dist, ties="topk")
        _, w_b = O.interpolate_from_mask(feats, fg, dist, ties="index")
        assert torch.equal(w_a, w_b)


def _run_oracle_case(case):
    q, k, v, mask, coords = case_inputs(case)
    cls = O.GeometryEditOracle if case["kind"] == "edit" else O.GeometryRemoverOracle
    c = cls(mask, cases.NUM_STEPS, cases.SELF_REPLACE, cases.OBJ_EDIT_STEP,
            coords_quant=torch.float16 if case["quant"] else None)
    c.amodal_mask = O.torch_erode(torch.from_numpy(cases.amodal_input(mask)))
    c.mask_new_warped = warped_mask(case["coords"])
    c.num_att_layers = 32
    c.cur_step = case["cur_step"]
    if case["cfg"]:
        c.coords_base, c.coords_edit, c.use_cfg = (2, 3), (3, 4), True
    else:
        c.coords_base, c.coords_edit, c.use_cfg = (0, 1), (1, 2), False
    grad = not case["cfg"]
    if grad:
        q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
    with torch.set_grad_enabled(grad):
        out = c(q, k, v, case["cross"], "up", transform_coords=coords, scale=case["D"] ** -0.5)
    return c, q, k, v, out


@pytest.mark.parametrize("name", list(cases.CONTROLLER_CASES))
def test_g6_controller_forward(name):
    case = cases.CONTROLLER_CASES[name]
    g = load("G6_" + name)
    c, q, k, v, out = _run_oracle_case(case)
    assert rel_err(out, g["out"]) < TOL
    assert (c.cur_att_layer, c.cur_step) == (int(g["cur_att_layer"]), int(g["cur_step"]))
    if not case["cfg"]:
        total = (out * case_gout(case, out.shape)).sum()
        if "loss" in g:
            assert abs(float(c.loss) - float(g["loss"])) <= TOL * max(1.0, abs(float(g["loss"])))
            kind = "cross" if case["cross"] else "self"
            for key, val in c.loss_log_dict[kind].items():
                assert abs(float(val) - float(g["log_" + key])) <= TOL * max(1.0, abs(float(g["log_" + key]))), key
            assert c.loss_log_dict["num_layers"] == int(g["num_layers"])
            total = total + c.loss
        dq, dk, dv = torch.autograd.grad(total, [q, k, v], allow_unused=True)
        assert rel_err(dq, g["dq"]) < 5 * TOL
        assert rel_err(dk if dk is not None else torch.zeros_like(k), g["dk"]) < 5 * TOL
        assert rel_err(dv if dv is not None else torch.zeros_like(v), g["dv"]) < 5 * TOL


def test_g7_counters():
    g = load("G7_counters")
    case = dict(cases.CONTROLLER_CASES["edit_self_late_16"])
    mask = cases.ellipse_mask()
    c = O.GeometryEditOracle(mask, cases.NUM_STEPS, cases.SELF_REPLACE, cases.OBJ_EDIT_STEP)
    c.amodal_mask = O.torch_erode(torch.from_numpy(cases.amodal_input(mask)))
    c.mask_new_warped = warped_mask("translate")
    c.num_att_layers, c.cur_step = 4, 46
    coords = torch.from_numpy(cases.make_coords("translate", mask))
    trace = []
    with torch.no_grad():
        for call in range(14):
            q, k, v = (torch.from_numpy(a) for a in cases.make_qkv(70 + call, 4, 1, 64, 64, 8))
            c(q, k, v, False, "mid", transform_coords=coords, scale=8 ** -0.5)
            if call == 7:
                c.cur_step -= 1
            trace.append((c.cur_att_layer, c.cur_step))
    assert np.array_equal(np.array(trace), g["trace"])


def test_g8_update_latent():
    g = load("G8_update_latent")
    rng = np.random.default_rng(8)
    lat = torch.from_numpy(rng.standard_normal((2, 4, 64, 64), dtype=np.float32))
    ctx = torch.from_numpy(rng.standard_normal((4, 77, 32), dtype=np.float32))
    wl = torch.from_numpy(rng.standard_normal((2, 4, 64, 64), dtype=np.float32))
    wc = torch.from_numpy(rng.standard_normal((4, 77, 32), dtype=np.float32))
    gl, gc = wl, 2 * ctx * wc * wc
    mask = torch.from_numpy(cases.ellipse_mask())[None, None]
    lo, co = O.update_latent(lat, gl, 0.37, mask[0, 0], ctx, gc)
    assert rel_err(lo, g["latents"]) < TOL
    assert rel_err(co, g["context"]) < TOL


def test_g9_adaptive():
    g = load("G9_adaptive")
    w_e = {"self": {"sim": 55, "movement": 30.5, "removal": 2.6, "smoothness": 30, "amodal": 80.5},
           "cross": {"sim": 45, "movement": 30.34, "removal": 2.6, "smoothness": 15, "amodal": 3.5}}
    w_r = {"self": {"sim": 55, "removal": 4.6, "smoothness": 30}, "cross": {"sim": 45, "removal": 4.6, "smoothness": 15}}
    for key, w, remover in (("edit", w_e, False), ("remover", w_r, True)):
        class C:
            pass
        c = C()
        c.default_loss_weights = {k: dict(v) for k, v in w.items()}
        c.loss_weight_dict = c.default_loss_weights
        c.initialize_default_loss_weights = lambda c=c: setattr(c, "loss_weight_dict", c.default_loss_weights)
        traj = []
        for i, val in g["seq"]:
            O.adaptive_step(c, int(i), 2, float(val), 50, -1.5, remover=remover)
            traj.append(c.loss_weight_dict["self"]["removal"])
        assert np.allclose(np.array(traj), g[key], rtol=1e-12, atol=0), key


def test_g10_ddim_closed_forms():
    g = load("G10_ddim")
    ac = O.alphas_cumprod()
    assert np.array_equal(ac.numpy(), g["alphas_cumprod"])
    rng = np.random.default_rng(10)
    x = torch.from_numpy(rng.standard_normal((1, 4, 8, 8), dtype=np.float32))
    e = torch.from_numpy(rng.standard_normal((1, 4, 8, 8), dtype=np.float32))
    prev = torch.stack([O.prev_step(e, int(t), x, ac, 50) for t in range(980, -1, -20)])
    nxt = torch.stack([O.next_step(e, int(t), x, ac, 50) for t in range(0, 1000, 20)])
    assert rel_err(prev, g["prev"]) < TOL
    assert rel_err(nxt, g["next"]) < TOL
    assert list(O.ddim_timesteps(50)) == list(range(980, -1, -20))
    # t = 0 inversion step is the identity (both alphas are alphas_cumprod[0])
    assert rel_err(nxt[0], x) < 1e-6


def test_g11_geometry_prepass():
    g = load("G11_geometry")
    size = 64
    mask = cases.ellipse_mask(cx=30, cy=33, ax=11, ay=9, size=size)
    v, u = np.mgrid[0:size, 0:size].astype(np.float32)
    depth = np.where(mask > 0.5, 0.5 + 0.2 * (u / size - 0.5), 0.9).astype(np.float32)
    for name in ("translate", "rotate", "mixed"):
        t, _ = O.get_transform_coordinates(depth.copy(), mask, torch.from_numpy(g[name + "_T"]).float(), focal_length=550 * size / 512.0)
        assert rel_err(t, g[name]) < 2e-5, name
    const = np.ones((size, size), dtype=np.float32) * 0.5
    T = torch.eye(4); T[0, 3] = 0.1
    t, _ = O.get_transform_coordinates(const, mask, T, focal_length=550 * size / 512.0)
    assert rel_err(t, g["const_depth"]) < 2e-5
    # the reference's documented example: tx = 0.1 on depth 0.5 is a pure shift of f*tx/Z pixels
    shift_px = (t[..., 0] - (2 * u / (size - 1) - 1)) * (size - 1) / 2
    assert np.allclose(shift_px, 550 * size / 512.0 * 0.1 / 0.5, atol=1e-3)


def test_g13_resample_and_morphology():
    g = load("G13_resample")
    mask = cases.ellipse_mask()
    coords = torch.from_numpy(cases.make_coords("rotate", mask))
    for S in (64, 8):
        assert np.array_equal(O.reshape_transform_coords(coords, S).numpy(), g[f"coords_{S}"])
        assert np.array_equal(O.reshape_attention_mask(torch.from_numpy(mask)[None, None], S).numpy(), g[f"mask_{S}"])
    blk = torch.zeros(1, 1, 9, 9); blk[..., 3:6, 3:6] = 1
    assert np.array_equal(O.torch_erode(blk, 3).numpy(), g["erode3"])
    assert np.array_equal(O.torch_dilate(blk, 3).numpy(), g["dilate3"])
    assert np.array_equal(O.torch_dilate(blk, 5).numpy(), g["dilate5"])
    assert rel_err(O.gaussian_kernel_5x5(), g["gauss_w"]) < 1e-6
    assert abs(float(O.gaussian_kernel_5x5()[2, 2]) - 0.18102) < 1e-4


@pytest.mark.parametrize("name", cases.HIST_CASES)
def test_g14_masked_histogram_matching(name):
    """Oracle == the reference's own float64 output, bit for bit (plateaus in the CDFs, soft masks, the identity-mask branch,
    one-pixel masks, constant images), plus the properties the reference's callers rely on."""
    g = load("G14_histogram")
    src, tmpl, m, ms = cases.hist_case(name)
    out = O.masked_histogram_matching(src, tmpl, m, ms)
    assert out.dtype == np.float64 and np.array_equal(out, g[name])
    for ch in range(3):                                   # a per-channel monotone look-up table
        order = np.argsort(src[..., ch].reshape(-1), kind="stable")
        assert np.all(np.diff(out[..., ch].reshape(-1)[order]) >= 0)
    if name.startswith("full_mask"):
        assert np.array_equal(O.masked_histogram_matching(src, src), src.astype(np.float64))     # self-match = identity


def test_g17_attention_store():
    """store_attention_maps: contents of attention_store after two 3-layer steps == the reference's (maps of the 16^2 layers only,
    list concatenation across steps, no length_ keys because cur_step is already 2 when the second merge happens)."""
    g = load("G17_attention_store")
    mask = cases.ellipse_mask()
    case = cases.CONTROLLER_CASES["edit_self_cfg_32"]
    c = O.GeometryEditOracle(mask, cases.NUM_STEPS, cases.SELF_REPLACE, cases.OBJ_EDIT_STEP, coords_quant=None)
    c.amodal_mask = O.torch_erode(torch.from_numpy(cases.amodal_input(mask)))
    c.mask_new_warped = warped_mask(case["coords"])
    c.num_att_layers, c.cur_step, c.store_attention_maps = 3, 0, True
    c.coords_base, c.coords_edit, c.use_cfg = (2, 3), (3, 4), True
    coords = torch.from_numpy(cases.make_coords(case["coords"], mask))
    with torch.no_grad():
        for step in range(2):
            for li, (S, cross, place) in enumerate(cases.STORE_LAYERS):
                q, k, v = (torch.from_numpy(a) for a in cases.make_qkv(900 + 10 * step + li, 4, 2, S * S, 77 if cross else S * S, 16))
                c(q, k, v, cross, place, transform_coords=coords, scale=0.25)
    assert c.cur_step == int(g["cur_step"])
    keys = [k[3:] for k in g if k.startswith("n__")]
    assert sorted(c.attention_store) == sorted(keys)
    for key in keys:
        assert len(c.attention_store[key]) == int(g["n__" + key])
        for i, a in enumerate(c.attention_store[key]):
            assert rel_err(a, torch.from_numpy(g[f"{key}__{i}"])) < 1e-5
    avg = c.get_average_attention()
    assert torch.equal(avg["down_self"][1], c.attention_store["down_self"][1] / 2)


def test_rasterizer_boxes_match_bruteforce():
    """The candidate-box rasterizer equals the every-pixel-visits-every-point form (small sizes)."""
    rng = np.random.default_rng(3)
    for S, P, K, rpx in ((16, 256, 4, 1.3), (24, 900, 15, 2.7), (8, 64, 2, 0.6)):
        pts = rng.uniform(-1.2, 1.2, size=(2, P, 3)).astype(np.float32)
        pts[..., 2] = np.round(rng.uniform(-0.1, 1.0, size=(2, P)) * 8) / 8      # many z ties + some z<0
        pts = torch.from_numpy(pts)
        a = O.rasterize_points(pts, S, rpx / S * 2.0, K)
        b = O.rasterize_points(pts, S, rpx / S * 2.0, K, bruteforce=True)
        for x, y in zip(a, b):
            assert torch.equal(x, y)


def test_g12_mesh_faces_and_vertices_match_the_reference():
    """get_mesh / create_triangles restated (oracle) vs the reference's own output (G12): face lists exact (per-TRIANGLE corner
    test, upper triangles first), vertices exact."""
    g = load("G12_mesh")
    for name, mask in cases.mesh_masks(64).items():
        v, f = O.get_mesh(g[name + "_t_coords"], mask)
        assert np.array_equal(f, g[name + "_faces"]), name
        assert np.array_equal(v, g[name + "_verts"]), name
    # the ellipse outline and the notches keep triangles whose 2x2 quad has its fourth corner outside the mask
    for name in ("ellipse", "holes"):
        m = cases.mesh_masks(64)[name] >= 0.5
        quads = (m[:-1, :-1] & m[:-1, 1:] & m[1:, :-1] & m[1:, 1:]).sum()
        assert g[name + "_faces"].shape[0] > 2 * quads, name


def test_mesh_coverage_oracle_bbox_walk_equals_bruteforce():
    """oracle/c/mesh_ref.c: the bounding-box walk accepts exactly the pixels of the all-pairs loop (rules M1-M5), including faces
    partly or wholly off screen, degenerate faces and an empty mesh."""
    g = load("G12_mesh")
    for name, mask in cases.mesh_masks(64).items():
        v, f = O.get_mesh(g[name + "_t_coords"], mask)
        for shift in (0.0, 0.9, 2.5):                       # on screen, half off screen, fully off screen
            vv = v.copy(); vv[:, 0] += shift
            a = O.splatter_mesh(vv, f, 64)
            assert np.array_equal(a, O.splatter_mesh(vv, f, 64, bruteforce=True)), (name, shift)
            if shift == 2.5:
                assert a.sum() == 0
            if shift == 0.0:
                assert a.sum() > 0
    # behind the camera (negative depth) -> nothing; degenerate (zero-area) faces -> nothing; empty mesh -> zeros
    v, f = O.get_mesh(g["ellipse_t_coords"], cases.mesh_masks(64)["ellipse"])
    vneg = v.copy(); vneg[:, 2] = -1.0
    assert O.splatter_mesh(vneg, f, 64).sum() == 0
    vdeg = v.copy(); vdeg[:, 1] = 0.25
    assert O.splatter_mesh(vdeg, f, 64).sum() == 0
    assert O.splatter_mesh(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32), 64).sum() == 0


@pytest.mark.parametrize("kind,fixture,cfgname", [("geometry_editor", "G18_loop", "LOOP"), ("geometry_remover", "G19_loop_remover", "LOOP"),
                                                   ("geometry_editor", "G20_loop_cfg0", "LOOP_CFG0")])
def test_oracle_loop_matches_the_reference_driver(kind, fixture, cfgname):
    """oracle/ref_loop.py (the per-step edit loop restated: optimisation pass -> latent update -> adaptive schedule -> CFG pass ->
    trajectory replacement -> latent warp, with the processor protocol) vs the REFERENCE's own ``text2image_ldm_stable`` recorded in
    G18 / G19 / G20 (same seeded narrow UNet, fp32, CPU): final latents, the loss log of every optimisation step, the first latent
    update and the final adaptive weight.  G20 is BASELINE configs[0] (256 x 256, 2-D translation, 20-step DDIM)."""
    import ref_loop
    from geodiffuser_amd.pipeline import build_random_sd21
    g = load(fixture)
    c = getattr(cases, cfgname)
    if cfgname == "LOOP_CFG0" and os.environ.get("GD_SLOW_TESTS", "1") != "1":
        pytest.skip("20-step loop skipped (GD_SLOW_TESTS=0)")
    torch.set_num_threads(8)
    pipe = build_random_sd21(device="cpu", dtype=torch.float32, tiny=True)
    probe = torch.cat([p.detach().reshape(-1)[:64] for p in pipe.unet.parameters()])
    if not torch.allclose(probe, torch.from_numpy(g["weight_probe"]), atol=1e-6):
        fixture_mismatch("seeded weights differ from the fixture's (different torch build)")
    inp = cases.loop_inputs(c)
    ctrl = ref_loop.make_controller(kind, inp["mask"], c, cases.amodal_input(inp["mask"], *c.get("amodal_shift", (32, -12))))
    tok = pipe.tokenizer
    ids = tok(["", ""], padding="max_length", max_length=tok.model_max_length, return_tensors="pt").input_ids
    with torch.no_grad():
        emb = pipe.text_encoder(ids)[0]
    updates = []
    orig = O.update_latent

    def rec(latents, g_lat, step, mask, ctx, g_ctx):
        res = orig(latents, g_lat, step, mask, ctx, g_ctx)
        updates.append((res[0][-1:].detach() - latents[-1:].detach()).clone())
        return res

    O.update_latent = rec
    try:
        lat, logs = ref_loop.text2image_loop(
            pipe.unet, emb, emb, ctrl, torch.from_numpy(inp["x_T"]), [torch.from_numpy(a) for a in inp["ddim_latents"]],
            torch.from_numpy(inp["coords"]), torch.from_numpy(inp["mask"]), num_steps=c["steps"], guidance_scale=c["guidance"],
            skip_optim_steps=c["skip_optim"], optimize_steps=c["optimize_steps"], latent_replace=c["latent_replace"], lr=c["lr"], edit_type=kind)
    finally:
        O.update_latent = orig
    assert sorted(logs) == list(g["steps"])
    for i, d in logs.items():
        for att in ("self", "cross"):
            for k, v in d[att].items():
                ref = float(g[f"log_{i}_{att}_{k}"])
                # first optimisation pass: identical inputs -> fp32 round-off only; later passes: that round-off amplified by the loop
                # (the reference itself moves by 4e-4 when x_T is perturbed by 1e-6, tests/golden/fp16_emulation.json)
                tol = 2e-4 if i == int(g["steps"][0]) else 2e-3
                assert abs(v - ref) <= tol * abs(ref) + 1e-6, (i, att, k, v, ref)
        assert d["num_layers"] == int(g[f"log_{i}_num_layers"])
    assert float(ctrl.loss_weight_dict["self"]["removal"]) == pytest.approx(float(g["final_weights_self_removal"]), rel=1e-9)
    # the L1 losses differentiate to sign(x): fp32 round-off between two restatements of the same arithmetic flips the unit gradient
    # of elements whose |edit_out - replace_out| is at round-off level (far background of the first pass), hence L2, not max-norm
    from _util import rel_l2
    e_up, e_lat = rel_l2(updates[0], torch.from_numpy(g["first_update"])), rel_l2(lat[-1:], torch.from_numpy(g["latents"])[-1:])
    print(f"[oracle loop] {fixture}: first update rel_l2 {e_up:.2e}, final latent rel_l2 {e_lat:.2e}")
    assert e_up < 2e-2 and e_lat < 5e-3


def test_quick_loop_fixtures_regenerate_bit_identically():
    """`oracle/gen_golden.py --check`: the loop fixtures that take < 60 s are re-recorded from the reference's own driver (imported from
    /root/reference: build container only) and must equal the committed files to the bit.  A long loop is reproducible only on the torch
    build / thread count named in its `provenance` entry (fp32 BLAS partitions: VERDICT r04 weak #3), which is why every fixture carries
    one and the generator pins its thread count."""
    import subprocess
    import sys
    if not os.path.isdir("/root/reference/GeoDiffuser"):
        pytest.skip("the reference tree is not on this machine (GPU box): fixtures are checked where they are recorded")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "oracle", "gen_golden.py"), "--check", "G18_loop", "G19_loop_remover", "G23_loop_sd14"],
                       capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("CHECK")]
    print("\n".join(lines))
    assert r.returncode == 0 and len(lines) == 3 and all("bit-identical" in ln for ln in lines), r.stdout[-2000:] + r.stderr[-2000:]


def test_loop_fixtures_name_their_environment():
    """Every loop fixture records where it was made (torch build, thread count, parallel backend hash, git revision)."""
    import json
    for name in ("G18_loop", "G19_loop_remover", "G20_loop_cfg0", "G21_loop_cfg0_full", "G22_loop_cfg1_full", "G23_loop_sd14",
                 "G26_loop_remover_full", "G27_loop_sdxl", "G28_loop_cfg1_t50", "G29_loop_remover768_t75", "G30_loop_cfg1_full_t50"):
        prov = json.loads(str(load(name)["provenance"]))
        assert {"torch", "threads", "parallel_info_md5", "git"} <= set(prov), name
        assert prov["threads"] == 8, (name, prov)

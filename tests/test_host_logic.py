"""CPU tests of the host-side mirror of the reference interface (no kernels are launched):
adaptive schedule (golden G9), scheduler tables, UNet harness topology / processor registration, loss-log handling,
multi-process sharding + weight broadcast over gloo (world size 2)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import ref_cpu as O
from _util import load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_adaptive_schedule_matches_golden():
    from geodiffuser_amd.optimization import adaptive_optimization_step_editing, adaptive_optimization_step_remover
    g = load("G9_adaptive")
    w_e = {"self": {"sim": 55, "movement": 30.5, "removal": 2.6, "smoothness": 30, "amodal": 80.5},
           "cross": {"sim": 45, "movement": 30.34, "removal": 2.6, "smoothness": 15, "amodal": 3.5}}
    w_r = {"self": {"sim": 55, "removal": 4.6, "smoothness": 30}, "cross": {"sim": 45, "removal": 4.6, "smoothness": 15}}
    for key, w, fn in (("edit", w_e, adaptive_optimization_step_editing), ("remover", w_r, adaptive_optimization_step_remover)):
        class C:
            pass
        c = C()
        c.default_loss_weights = {k: dict(v) for k, v in w.items()}
        c.loss_weight_dict = c.default_loss_weights
        c.initialize_default_loss_weights = lambda c=c: setattr(c, "loss_weight_dict", c.default_loss_weights)
        traj = []
        for i, val in g["seq"]:
            fn(c, int(i), 2, {"self": {"removal": float(val)}}, num_ddim_steps=50, removal_loss_value_in=-1.5)
            traj.append(c.loss_weight_dict["self"]["removal"])
        assert np.allclose(np.array(traj), g[key], rtol=1e-12, atol=0), key


def test_scheduler_tables():
    from geodiffuser_amd.scheduler import DDIMInverseScheduler, DDIMScheduler
    g = load("G10_ddim")
    s = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False)
    assert np.array_equal(s.alphas_cumprod.numpy(), g["alphas_cumprod"])
    s.set_timesteps(50)
    assert s.timesteps.tolist() == list(range(980, -1, -20))
    assert float(s.final_alpha_cumprod) == float(g["alphas_cumprod"][0])
    inv = DDIMInverseScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False)
    inv.set_timesteps(50)
    assert inv.timesteps.tolist() == list(range(0, 1000, 20))
    s.set_timesteps(20)
    assert s.timesteps.tolist() == list(range(950, -1, -50))
    with pytest.raises(NotImplementedError):
        DDIMScheduler(clip_sample=True)


def test_unet_harness_topology_and_registration():
    from geodiffuser_amd.attention_processors import (AttentionGeometryEdit, EditProcessor, VanillaAttentionProcessor,
                                                      register_attention_control_diffusers, set_attn_processor_for_edit)
    from geodiffuser_amd.unet_sd21 import UNet2DConditionModel
    with torch.device("meta"):
        unet = UNet2DConditionModel()
    assert sum(p.numel() for p in unet.parameters()) == 865910724          # public SD2.1-base UNet parameter count
    names = list(unet.attn_processors)
    assert len(names) == 32
    assert names[0] == "down_blocks.0.attentions.0.transformer_blocks.0.attn1.processor"
    assert sum(n.startswith("down_blocks") for n in names) == 12 and sum(n.startswith("mid_block") for n in names) == 2
    heads = {n: m.heads for n, m in unet.named_modules() if hasattr(m, "heads")}
    assert set(heads.values()) == {5, 10, 20}
    assert all(abs(m.scale - 0.125) < 1e-12 for m in unet.modules() if hasattr(m, "scale"))

    class Model:
        pass
    model = Model(); model.unet = unet
    mask = np.zeros((512, 512), np.float32); mask[200:300, 200:300] = 1
    ctrl = AttentionGeometryEdit(["", ""], 50, {"default_": 0.95}, 0.95, image_mask=mask, obj_edit_step=0.9)
    register_attention_control_diffusers(model, ctrl, transform_coords=None)
    assert ctrl.num_att_layers == 32
    procs = unet.attn_processors
    assert all(isinstance(p, EditProcessor) for p in procs.values())
    places = [p.place_in_unet for p in procs.values()]
    assert places.count("down") == 12 and places.count("mid") == 2 and places.count("up") == 18
    set_attn_processor_for_edit(model, coords_base=(0, 1), coords_edit=(1, 2), use_cfg=False)
    assert ctrl.coords_base == (0, 1) and ctrl.coords_edit == (1, 2) and ctrl.use_cfg is False
    assert ctrl.num_self_replace == (0, 47) and type(ctrl).__name__ == "AttentionGeometryEdit"
    # aliasing of the default weights (reference behaviour)
    ctrl.loss_weight_dict["self"]["removal"] *= 2
    ctrl.initialize_default_loss_weights()
    assert ctrl.loss_weight_dict["self"]["removal"] == 1.67 * 2
    unet.set_attn_processor(VanillaAttentionProcessor())
    assert all(isinstance(p, VanillaAttentionProcessor) for p in unet.attn_processors.values())


def test_loss_log_helpers_and_errors():
    from geodiffuser_amd import editor
    from geodiffuser_amd.attention_processors import AttentionGeometryRemover
    mask = np.zeros((512, 512), np.float32); mask[100:140, 100:140] = 1
    c = AttentionGeometryRemover(["", ""], 50, {"default_": 0.9}, 0.9, image_mask=mask, obj_edit_step=1.0)
    assert float(c.image_mask[0].sum()) == 44 * 44                           # 5x5 dilation at construction (:986)
    assert set(c.loss_log_dict["self"]) == {"sim", "removal", "smoothness"}
    c.loss_log_dict["self"]["sim"] = torch.tensor(1.5)
    out = editor.convert_loss_log_to_numpy(c.loss_log_dict)
    assert out["self"]["sim"] == 1.5 and out["cross"]["removal"] == 0.0 and out["num_layers"] == 0
    c.loss = torch.tensor(3.0)
    editor.clear_controller_loss(c)
    assert c.loss == 0.0 and c.loss_log_dict["self"]["sim"] == 0.0
    with pytest.raises(NameError):                                            # same failure as the reference (editor.py:618-621)
        editor.perform_geometric_edit(np.zeros((8, 8, 3), np.uint8), np.ones((8, 8), np.float32), np.ones((8, 8), np.float32),
                                      torch.eye(4), edit_type="geometry_stitch")


_WORKER = r'''
import os, sys, torch
sys.path.insert(0, %r)
from geodiffuser_amd import dist as gd
rank, world, local = gd.init(backend="gloo")
assert world == 2
torch.manual_seed(100 + rank)                       # different weights per rank before the broadcast
m = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Linear(32, 8)).half()
sent = gd.broadcast_model([m], src=0, bucket_bytes=256)
torch.manual_seed(100)
ref = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Linear(32, 8)).half()
for a, b in zip(m.parameters(), ref.parameters()):
    assert torch.equal(a, b)
mine = gd.shard(list(range(7)), rank, world)
assert mine == [j for j in range(7) if j %% 2 == rank]
t = gd.max_over_ranks(1.0 + rank)
n = gd.sum_over_ranks(float(len(mine)))
gd.barrier()
assert t == 2.0 and n == 7.0 and sent > 0
print("rank", rank, "ok")
'''


def test_two_process_gloo_shard_and_broadcast(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29612", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o


def test_diffusion_step_with_a_foreign_scheduler():
    """a scheduler object that is not the repo's own (e.g. a diffusers DDIMScheduler passed as scheduler_in) has
    the plain ``step(model_output, t, sample, eta=0.0)`` signature: diffusion_step must do the classifier-free-guidance combine itself
    (diffusion.py:47-49) instead of passing the fused-kernel keywords."""
    import torch
    from types import SimpleNamespace
    from geodiffuser_amd.diffusion import diffusion_step

    class ForeignScheduler:
        def step(self, model_output, timestep, sample, eta=0.0):           # no **kwargs, like diffusers 0.25.1
            return {"prev_sample": sample - 0.5 * model_output}

    class FakeUNet:
        def __call__(self, x, t, encoder_hidden_states=None):
            return {"sample": x * 2.0 + encoder_hidden_states.mean(dim=(1, 2))[:, None, None, None]}

    class Ctrl:
        def step_callback(self, x, coords=None):
            return x

    model = SimpleNamespace(unet=FakeUNet(), scheduler=ForeignScheduler())
    torch.manual_seed(0)
    lat = torch.randn(2, 4, 8, 8)
    ctx = torch.randn(4, 77, 16)
    g = 3.0
    eps = model.unet(torch.cat([lat] * 2), 500, encoder_hidden_states=ctx)["sample"]
    eu, ec = eps.chunk(2)
    want = lat - 0.5 * (eu + g * (ec - eu))
    got = diffusion_step(model, Ctrl(), lat, ctx, 500, g)
    assert torch.allclose(got, want, atol=1e-6)
    got2, noise = diffusion_step(model, Ctrl(), lat, ctx, 500, g, return_noise=True)
    assert torch.allclose(got2, want, atol=1e-6) and torch.allclose(noise, eu + g * (ec - eu), atol=1e-6)
    # 3-row shortcut (uncond_edit, cond_ref, cond_edit)
    got3 = diffusion_step(model, Ctrl(), lat, ctx, 500, g, skip_uncond_ref=True)
    assert torch.allclose(got3[1:], want[1:], atol=1e-6) and torch.equal(got3[:1], lat[:1])


def test_sdxl_harness_topology():
    """The SDXL-base-shaped UNet harness (BASELINE configs[4]): the public model's parameter count and attention-module census
    (70 transformer blocks x {attn1, attn2}), built on the meta device (no memory)."""
    from geodiffuser_amd.unet_sd21 import sdxl_unet
    with torch.device("meta"):
        u = sdxl_unet()
    assert sum(p.numel() for p in u.parameters()) == 2_567_463_684
    names = list(u.attn_processors)
    assert len(names) == 140 and not any(n.startswith("down_blocks.0") or n.startswith("up_blocks.2") for n in names)
    heads = {m.heads for _, m in u._attn_modules()}
    assert heads == {10, 20}


def test_sd14_harness_topology():
    """The SD1.x-shaped harness (the reference's default model, U/editor.py:58): CompVis/stable-diffusion-v1-4's UNet parameter count and
    its 8-head layout (head dims 40 / 80 / 160), on the meta device."""
    from geodiffuser_amd.unet_sd21 import sd14_unet
    with torch.device("meta"):
        u = sd14_unet()
    assert sum(p.numel() for p in u.parameters()) == 859_520_964
    assert sorted({m.to_q.out_features // m.heads for _, m in u._attn_modules()}) == [40, 80, 160]
    assert {m.heads for _, m in u._attn_modules()} == {8} and len(u.attn_processors) == 32


def test_batched_qkv_projection_equals_separate_linears():
    """The one-GEMM projection of the no-grad passes (stride-0 broadcast input, stacked weights, the query's scale*log2(e) folded into its
    weight) against the three / two separate linears, on CPU in fp32; the weight cache follows in-place weight updates."""
    from geodiffuser_amd import attention_processors as AP
    from geodiffuser_amd.unet_sd21 import Attention
    torch.manual_seed(0)
    for ctx_dim in (None, 24):
        attn = Attention(32, ctx_dim, heads=2, dim_head=16)
        x = torch.randn(2, 10, 32)
        ctx = None if ctx_dim is None else torch.randn(2, 7, ctx_dim)
        q, k, v, is_cross, _, _, q_scaled = AP._batched_qkv(attn, x, ctx)
        src = x if ctx is None else ctx
        assert q_scaled and is_cross == (ctx is not None)
        assert torch.allclose(q, attn.to_q(x) * (attn.scale * AP.LOG2E), atol=1e-5)
        assert torch.allclose(k, attn.to_k(src), atol=1e-5) and torch.allclose(v, attn.to_v(src), atol=1e-5)
        assert q.is_contiguous() and k.is_contiguous() and v.is_contiguous()
        with torch.no_grad():
            attn.to_k.weight.mul_(2.0)                             # the stack is rebuilt when a weight changes
        k2 = AP._batched_qkv(attn, x, ctx)[1]
        assert torch.allclose(k2, attn.to_k(src), atol=1e-5) and not torch.allclose(k2, k, atol=1e-3)
    attn.to_q.bias = torch.nn.Parameter(torch.zeros(32))          # projections with a bias do not qualify
    assert AP._batched_qkv(attn, x, ctx) is None


def test_bench_gpus_n_starts_its_own_ranks(tmp_path, monkeypatch):
    """VERDICT r02 missing #2: ``python bench.py --gpus N`` with no torchrun environment must start N ranks itself (as a child process,
    before the parent touches the GPU), fail loudly when fewer than N devices are visible, and refuse a WORLD_SIZE that contradicts
    --gpus.  The launcher branch is exercised here with an injected device count / runner; the N-rank data path is covered by the
    two-process gloo tests."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    # fewer devices than ranks: exit code 2, nothing started
    started = []
    assert bench.spawn_ranks(8, ["--gpus", "8"], device_count=1, run=lambda cmd, env: started.append(cmd) or 0) == 2
    assert not started
    # enough devices: one torchrun child with N ranks, rendezvous on 127.0.0.1 at a free port, the original arguments forwarded
    rc = bench.spawn_ranks(2, ["--gpus", "2", "--steps", "3"], device_count=2, run=lambda cmd, env: started.append((cmd, env)) or 7)
    assert rc == 7 and len(started) == 1
    cmd, env = started[0]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # the real entry point on this GPU-less box: --gpus 2 must fail loudly (no silent single-process run)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    if not torch.cuda.is_available():
        assert p.returncode == 2 and "only 0 GPU(s) visible" in p.stderr and not p.stdout.strip()


def test_bench_gpus_2_dry_run_really_starts_two_ranks():
    """the `--gpus N` launcher's first contact, end to end and for real — ``python bench.py --gpus 2 --dry-run`` starts
    ``python -m torch.distributed.run`` as a child, two rank processes rendezvous on 127.0.0.1 at the free port the launcher picked (gloo:
    no GPU here), rank 0's weights are broadcast in buckets (each rank built DIFFERENT weights), every rank reports one stderr line,
    and ONLY rank 0 prints the JSON line, with one entry per rank in the gathered fields.  The edit itself is a stub in this mode (the
    hot path has no CPU implementation) and the line says so: it can never pass for a measurement."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                                   # rank 0 only
    line = json.loads(lines[0])
    assert line["dry_run"] is True and line["value"] is None and line["n_gpus"] == 2 and "DRY RUN" in line["metric"]
    cfg = line["config"]
    assert cfg["weights_broadcast_bytes"] > 0 and cfg["weights_equal_after_broadcast"] is True
    assert len(cfg["per_rank_s"]) == 2 and len(cfg["first_warmup_edit_s"]) == 2
    assert cfg["edits_by_rank"] == {"0": [0, 2, 4], "1": [1, 3, 5]}      # edit j -> rank j mod W, every edit exactly once
    for r in (0, 1):
        assert f"[bench rank {r}/2] device cpu (dry run)" in p.stderr
    # every rank of a node on its own slice of the host cores (dist.pin_rank_to_cores, set in-process before anything else starts threads)
    if hasattr(os, "sched_getaffinity") and len(os.sched_getaffinity(0)) >= 2:
        (a0, a1, na), (b0, b1, nb) = cfg["host_cores_by_rank"]
        assert na >= 1 and nb >= 1 and a1 < b0, cfg["host_cores_by_rank"]


def test_pin_rank_to_cores_partitions_the_visible_cores():
    """dist.pin_rank_to_cores in a child process per rank (the mask is per process): contiguous, disjoint, equal slices of the cores the
    process may use; torch's intra-op pool sized to the slice; nothing happens for a single rank per node or with GD_PIN_CORES=0."""
    if not hasattr(os, "sched_getaffinity") or len(os.sched_getaffinity(0)) < 4:
        pytest.skip("needs sched_setaffinity and >= 4 visible cores")
    code = ("import os, json, torch; from geodiffuser_amd import dist; c = dist.pin_rank_to_cores(); "
            "print(json.dumps([c, sorted(os.sched_getaffinity(0)), torch.get_num_threads()]))")
    import json
    allc = sorted(os.sched_getaffinity(0))
    got = []
    for r in range(4):
        env = dict(os.environ, LOCAL_RANK=str(r), LOCAL_WORLD_SIZE="4", PYTHONPATH=ROOT)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-1000:]
        c, mask, nt = json.loads(out.stdout.strip().splitlines()[-1])
        assert c == mask and 1 <= nt <= len(c)
        got.append(c)
    per = len(allc) // 4
    assert [x for c in got for x in c] == allc[:4 * per] and all(len(c) == per for c in got)
    for extra in (dict(LOCAL_WORLD_SIZE="1"), dict(LOCAL_WORLD_SIZE="4", GD_PIN_CORES="0")):
        env = dict(os.environ, LOCAL_RANK="0", PYTHONPATH=ROOT, **extra)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        c, mask, _ = json.loads(out.stdout.strip().splitlines()[-1])
        assert c is None and mask == allc


def test_edits_in_flight_starts_p_ranks_per_gpu():
    """VERDICT r03 #7 (the batch driver's throughput mode): ``--edits-in-flight P`` starts P ranks per GPU — rank r drives device
    LOCAL_RANK // P, the control plane runs over gloo (RCCL takes one rank per device) with the weight broadcast staged through the host,
    and the line reports the GPU count, not the rank count.  Dry run on CPU: launcher, rendezvous, broadcast, sharding and reporting."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "GD_EDITS_IN_FLIGHT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--edits-in-flight", "3", "--steps", "2", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    cfg = line["config"]
    assert line["n_gpus"] == 1 and cfg["edits_in_flight_per_gpu"] == 3 and cfg["device_of_rank"] == [0, 0, 0]
    assert len(cfg["per_rank_s"]) == 3 and cfg["weights_equal_after_broadcast"] is True
    assert cfg["edits_by_rank"] == {"0": [0, 3], "1": [1, 4], "2": [2, 5]}
    from geodiffuser_amd import dist as gdist
    os.environ["GD_EDITS_IN_FLIGHT"] = "2"
    try:
        assert gdist.procs_per_gpu() == 2 and [gdist.local_device_index(r) for r in range(6)] == [0, 0, 1, 1, 2, 2]
    finally:
        del os.environ["GD_EDITS_IN_FLIGHT"]
    assert gdist.procs_per_gpu() == 1 and gdist.local_device_index(5) == 5


def test_visible_gpu_count_reads_masks_without_hip(monkeypatch):
    """The launcher counts devices from the visibility masks / the kernel driver's topology, never through the HIP runtime."""
    from geodiffuser_amd import dist
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    base = dist.visible_gpu_count()
    assert base >= -1
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert dist.visible_gpu_count() == (3 if base < 0 else min(3, base))
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0")
    assert dist.visible_gpu_count() == (1 if base < 0 else min(1, base))
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert dist.visible_gpu_count() == 0


def test_text_embeddings_are_cached_per_token_ids():
    """diffusion.encode_text: the four text-encoder calls of an edit (U/editor.py:116-121, U/inversion.py:213-224) hit a per-encoder cache
    keyed by the token ids; the cached value equals a direct call, callers get copies, an in-place weight edit drops the cache."""
    from types import SimpleNamespace
    from geodiffuser_amd import diffusion
    from geodiffuser_amd.pipeline import build_random_sd21
    pipe = build_random_sd21(device="cpu", dtype=torch.float32, tiny=True)
    model = SimpleNamespace(text_encoder=pipe.text_encoder, device=torch.device("cpu"))
    calls = []
    h = pipe.text_encoder.register_forward_hook(lambda *a: calls.append(1))
    tok = pipe.tokenizer
    ids = tok([""], padding="max_length", max_length=tok.model_max_length, return_tensors="pt").input_ids
    ids2 = tok(["a photo"], padding="max_length", max_length=tok.model_max_length, return_tensors="pt").input_ids
    try:
        e1 = diffusion.encode_text(model, ids)
        e2 = diffusion.encode_text(model, ids.clone())
        assert len(calls) == 1 and torch.equal(e1, e2) and e1.data_ptr() != e2.data_ptr()
        assert torch.equal(e1, pipe.text_encoder(ids)[0])
        n = len(calls)
        e3 = diffusion.encode_text(model, ids2)
        assert len(calls) == n + 1 and not torch.equal(e3, e1)
        e1.zero_()                                                   # a caller's copy is its own
        assert torch.equal(diffusion.encode_text(model, ids), e2)
        with torch.no_grad():
            next(pipe.text_encoder.parameters()).mul_(1.5)           # weights changed in place: recomputed
        n = len(calls)
        e4 = diffusion.encode_text(model, ids)
        assert len(calls) == n + 1 and not torch.equal(e4, e2)
    finally:
        h.remove()


def test_miopen_cache_works_on_a_scratch_copy(tmp_path, monkeypatch):
    """the committed find-db is a read-only seed (scratch copy per process), GD_MIOPEN_CACHE=0 is honoured everywhere, and a
    db written under ANOTHER MIOpen build name is reported."""
    import importlib
    import warnings
    from geodiffuser_amd import miopen_cache as mc
    for k in ("MIOPEN_USER_DB_PATH", "MIOPEN_CUSTOM_CACHE_DIR", "GD_MIOPEN_DB", "GD_MIOPEN_DB_RECORD"):
        monkeypatch.delenv(k, raising=False)
    mc = importlib.reload(mc)
    monkeypatch.setenv("GD_MIOPEN_CACHE", "0")
    assert mc.configure() == "" and "MIOPEN_USER_DB_PATH" not in os.environ
    monkeypatch.setenv("GD_MIOPEN_CACHE", "1")
    seed = sorted(os.listdir(mc._DIR))
    d = mc.configure()
    try:
        assert os.path.realpath(d) != os.path.realpath(mc._DIR) and os.environ["MIOPEN_USER_DB_PATH"] == d
        assert mc.configure() == d                                           # idempotent
        assert [n for n in mc._db_names(d)] == [n for n in seed if n.endswith(".ufdb.txt")]
        assert mc.check_db_used() is True
        open(os.path.join(d, "gfx950100.HIP.9_9_9_other-build.ufdb.txt"), "w").write("x")
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            assert mc.check_db_used() is False
        assert any("no record for the running MIOpen build" in str(x.message) for x in w)
        assert sorted(os.listdir(mc._DIR)) == seed                           # the package directory is untouched
    finally:
        mc._cleanup(d)
        assert not os.path.exists(d)
        monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False)
        monkeypatch.delenv("MIOPEN_CUSTOM_CACHE_DIR", raising=False)
        importlib.reload(mc)


def test_dist_init_needs_a_port_from_the_launcher(monkeypatch):
    """More than one rank without MASTER_PORT: a loud error, not a silent 29500 (the ranks cannot agree on a port among themselves)."""
    from geodiffuser_amd import dist
    monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("RANK", "0"); monkeypatch.delenv("MASTER_PORT", raising=False)
    monkeypatch.setenv("GD_PIN_CORES", "0")          # (this is the test process itself: it must keep all of its cores)
    before = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None
    with pytest.raises(RuntimeError, match="MASTER_PORT"):
        dist.init("gloo")
    assert before is None or sorted(os.sched_getaffinity(0)) == before


def test_batch_driver_starts_its_own_ranks(monkeypatch):
    """``python -m geodiffuser_amd.large_scale_editor --gpus N --edits-in-flight P``: the batch driver's own launcher — N x P ranks of the
    module through torch.distributed.run as a child, GD_EDITS_IN_FLIGHT exported, the launcher flags stripped from the ranks' arguments,
    a refusal when fewer devices are visible (counted without HIP)."""
    from geodiffuser_amd import dist as gdist, large_scale_editor as L
    calls = []
    monkeypatch.setattr(L.main, "_run", lambda cmd, env: calls.append((cmd, env)) or 0, raising=False)
    monkeypatch.setattr(gdist, "visible_gpu_count", lambda: 8)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as ex:
        L.main(["--root", "/data/x", "--gpus", "2", "--edits-in-flight", "4", "--dtype", "fp16"])
    assert ex.value.code == 0 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd and env["GD_EDITS_IN_FLIGHT"] == "4"
    i = cmd.index("geodiffuser_amd.large_scale_editor")
    assert cmd[i - 1] == "-m" and cmd[i + 1:] == ["--root", "/data/x", "--dtype", "fp16"]
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    monkeypatch.setattr(gdist, "visible_gpu_count", lambda: 1)
    with pytest.raises(SystemExit) as ex:
        L.main(["--root", "/data/x", "--gpus", "2"])
    assert ex.value.code == 2 and len(calls) == 1


def test_edit_batch_routes_role_major_rows_and_keeps_counters_in_step(monkeypatch):
    """geodiffuser_amd.batch.EditBatch on CPU with stand-in controllers: edit j of B sees rows j, B + j, 2 B + j of the role-major batch
    (its own [uncond_edit, cond_ref, cond_edit] rows) with its own transform coordinates, the outputs come back role-major, the batch's
    layer / step counters and every edit's advance together, the driver's `cur_step -= 1` reaches every edit, and the graph key / table
    signature / loss state the hipGraph code asks for are assembled per edit."""
    import geodiffuser_amd.batch as GB
    from geodiffuser_amd.attention_sharing import AttentionControl

    class Sub(AttentionControl):
        _is_remover = False
        rows_identical = False

        def __init__(self, tag):
            super().__init__()
            self.tag, self.seen = tag, []
            self.num_steps, self.obj_edit_step, self.num_self_replace = 50, 0.9, (0, 47)
            self.masks_cache_dict = {32: {"f": 4, "D": 64, "S": 32}}
            self.loss, self.loss_log_dict = 0.0, {"self": {"sim": 0.0}, "cross": {"sim": 0.0}, "num_layers": 0}

        def forward(self, q, k, v, is_cross, place_in_unet, transform_coords=None, scale=None, mask=None):
            self.seen.append((q[:, 0, 0].tolist(), float(transform_coords), self.coords_base, self.heads_tok))
            return q + self.tag

        def graph_key(self):
            return ("Sub", True, True, self.n_batch, self.coords_base, self.coords_edit, self.use_cfg, False, bool(self.rows_identical))

        def table_signature(self):
            return ((32, 256 + self.tag, 4, 15, 0),)

        def tables_built(self, layers):
            return True

        def after_graph_replay(self):
            self.cur_att_layer = 0
            self.cur_step += 1

    monkeypatch.setattr(GB, "MERGED", False)
    B = 3
    subs = [Sub(100 * (j + 1)) for j in range(B)]
    batch = GB.EditBatch(subs, [torch.tensor(float(j)) for j in range(B)])
    assert [s.slot for s in subs] == [0, 1, 2]
    batch.num_att_layers = 2
    batch.coords_base, batch.coords_edit, batch.use_cfg, batch.n_batch = (1, 2), (2, 3), True, 3
    q = torch.arange(9.0).view(9, 1, 1).expand(9, 4, 8).contiguous()          # row r of the role-major batch carries the value r
    for layer in range(2):
        batch.heads_tok = 4
        out = batch(q, q, q, False, "up", scale=0.125)
    assert [s.seen[0][0] for s in subs] == [[0.0, 3.0, 6.0], [1.0, 4.0, 7.0], [2.0, 5.0, 8.0]]
    assert [s.seen[0][1:] for s in subs] == [(float(j), (1, 2), 4) for j in range(B)]
    assert out[:, 0, 0].tolist() == [100.0, 201.0, 302.0, 103.0, 204.0, 305.0, 106.0, 207.0, 308.0]       # role-major again, edit j's tag on its rows
    assert (batch.cur_att_layer, batch.cur_step) == (0, 1) and all((s.cur_att_layer, s.cur_step) == (0, 1) for s in subs)
    batch.undo_step()
    assert batch.cur_step == 0 and all(s.cur_step == 0 for s in subs)
    # (last entry: the serial number of the optimisation-pass graph whose reference rows a "cfg2s" pass reads; None otherwise)
    assert batch.graph_key()[:2] == ("EditBatch", 3) and batch.graph_key()[-2:] == ((False, False, False), None)
    batch.use_ref_stash, batch.ref_stash_serial = True, 7
    assert batch.graph_key()[-1] == 7
    batch.use_ref_stash = False
    assert batch.table_signature() == tuple(s.table_signature() for s in subs)
    subs[1].loss_log_dict["self"]["sim"] = 7.0
    st = batch.export_loss_state()
    subs[1].loss_log_dict["self"]["sim"] = 0.0
    batch.import_loss_state(st)
    assert subs[1].loss_log_dict["self"]["sim"] == 7.0 and st[1][1] is not subs[1].loss_log_dict
    batch.after_graph_replay()
    assert batch.cur_step == 1 and all(s.cur_step == 1 for s in subs)
    with pytest.raises(ValueError):
        GB.EditBatch([Sub(1), type("Other", (Sub,), {})(2)], [torch.tensor(0.0), torch.tensor(0.0)])


def test_batch_entry_point_refuses_what_it_does_not_implement():
    """perform_geometric_edit_batch takes perform_geometric_edit's keyword arguments; what only the one-edit driver implements is refused
    loudly (before anything touches the device), unknown names are a TypeError like any Python call."""
    from geodiffuser_amd.batch import perform_geometric_edit_batch
    with pytest.raises(TypeError):
        perform_geometric_edit_batch([], no_such_argument=1)
    for kw in (dict(fast_start_steps=0.2), dict(return_attention_maps=True), dict(perform_inversion=True), dict(edit_type="geometry_stitch")):
        with pytest.raises(NotImplementedError):
            perform_geometric_edit_batch([], **kw)
    assert perform_geometric_edit_batch([], progress=None, use_optimizer=True, num_first_optim_steps=1) == []


def test_start_ahead_runs_in_line_without_a_gpu_and_reports_errors(monkeypatch):
    """editor.start_ahead: on a CPU device (or GD_PREPASS_THREAD=0) the body runs on the caller's thread at the call; its value comes
    back through result(); an exception in the body is raised, not swallowed."""
    import threading
    from geodiffuser_amd import editor
    monkeypatch.setattr(editor, "DEVICE", torch.device("cpu"))
    seen = []
    h = editor.start_ahead(lambda: seen.append(threading.current_thread().name) or 41 + 1)
    assert seen == [threading.current_thread().name] and h.result() == 42
    with pytest.raises(ZeroDivisionError):
        editor.start_ahead(lambda: 1 / 0).result()


def test_batch_entry_point_runs_long_lists_in_chunks(monkeypatch):
    """More edits than one launch has per-edit segments (batch.MAX_EDITS): the list runs MAX_EDITS at a time, results in list order."""
    import geodiffuser_amd.batch as GB
    orig, calls = GB.perform_geometric_edit_batch, []

    def spy(edits, **kw):
        calls.append((len(edits), kw.get("edit_type"), kw.get("guidance_scale")))
        if len(edits) > GB.MAX_EDITS:
            return orig(edits, **kw)
        return [("result", e["id"]) for e in edits]

    monkeypatch.setattr(GB, "perform_geometric_edit_batch", spy)
    n = 2 * GB.MAX_EDITS + 3
    res = spy([{"id": i} for i in range(n)], edit_type="geometry_remover", guidance_scale=5.0)
    assert [c[0] for c in calls] == [n, GB.MAX_EDITS, GB.MAX_EDITS, 3] and all(c[1:] == ("geometry_remover", 5.0) for c in calls)
    assert res == [("result", i) for i in range(n)]
    # a hooked layer of a batch of B edits launches per group of GROUP: [0, 8), [8, B)
    assert GB._groups(5) == [(0, 5)] and GB._groups(GB.GROUP) == [(0, GB.GROUP)] and GB._groups(11) == [(0, 8), (8, 11)] and GB.MAX_EDITS == 2 * GB.GROUP

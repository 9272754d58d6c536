"""N3 — experiment-folder wire format, transform composition and the batch driver's configuration table / work list,
against fixtures written by the reference itself (tests/golden/exp_root/ by its save_exp; G15 = what its read_exp returns for
them and what get_transformed_mask composes; G16 = what its perform_exp passes to perform_geometric_edit)."""
import json
import os

import numpy as np
import pytest
import torch

import cases
from _util import GOLDEN as GOLDEN_DIR, load

ROOT = os.path.join(GOLDEN_DIR, "exp_root")


@pytest.mark.parametrize("cat,idx", cases.EXP_CASES)
def test_read_exp_matches_reference(cat, idx):
    from geodiffuser_amd.ui_utils import read_exp
    g = load("G15_exp_folder")
    d = read_exp(os.path.join(ROOT, cat, str(idx)))
    none_keys = sorted(k for k, v in d.items() if v is None)
    assert none_keys == list(g[f"{cat}_{idx}__none_keys"])
    n = 0
    for k, v in d.items():
        if isinstance(v, np.ndarray):
            ref = g[f"{cat}_{idx}__{k}"]
            assert v.dtype == ref.dtype and v.shape == ref.shape and np.array_equal(v, ref), k
            n += 1
    assert n >= 6 and d["path_name"].endswith(os.path.join(cat, str(idx)))
    assert d["input_image_png"].dtype == np.uint8 and d["input_image_png"].shape == (48, 64, 3)


def test_read_exp_defaults_and_missing(tmp_path):
    from geodiffuser_amd.ui_utils import read_exp
    d = read_exp(str(tmp_path))
    assert d["input_image_png"] is None and d["depth_npy"] is None
    assert np.array_equal(d["image_shape_npy"], np.array([512, 512]))              # ui_utils.py:156-157


@pytest.mark.parametrize("cat,idx", cases.EXP_CASES)
def test_save_exp_round_trip_equals_reference_written_folder(tmp_path, cat, idx):
    """A folder written by our save_exp reads back exactly like the one the reference wrote for the same arrays
    (data files exact; depth.png is a min/max-normalised visualisation — also exact here)."""
    from geodiffuser_amd.ui_utils import read_exp, save_exp
    g = load("G15_exp_folder")
    for i in range(1, idx):                                                       # numbering = existing folders + 1
        os.makedirs(tmp_path / cat / str(i))
    e = cases.exp_case(cat, idx)
    folder = save_exp(str(tmp_path), e["image"], e["depth"], e["depth_vis"], e["mask"], e["transform"],
                      transformed_image=e.get("transformed"), background_image=e.get("background"), h=e["h"], w=e["w"],
                      exp_transform_type=cat)
    assert folder.rstrip("/").endswith(os.path.join(cat, str(idx)))
    d = read_exp(folder)
    for k, v in d.items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(v, g[f"{cat}_{idx}__{k}"]), k
    assert np.array_equal(d["image_shape_npy"], [e["h"], e["w"]])


def test_check_if_exp_root_and_work_list():
    from geodiffuser_amd.large_scale_editor import list_experiments
    from geodiffuser_amd.ui_utils import check_if_exp_root
    from geodiffuser_amd.dist import shard
    g = load("G15_exp_folder")
    assert check_if_exp_root(ROOT) == bool(g["is_root"]) and check_if_exp_root(os.path.join(ROOT, "Mix")) == bool(g["is_root_leaf"])
    work = list_experiments(ROOT)
    rel = [(os.path.relpath(f, ROOT).rstrip("/"), t) for f, t in work]
    # Rotation_2D is skipped, Removal maps to the remover (large_scale_editor.py:376-388)
    assert rel == [("Mix/1", "geometry_editor"), ("Mix/2", "geometry_editor"), ("Removal/1", "geometry_remover")]
    assert shard(work, 0, 2) == [work[0], work[2]] and shard(work, 1, 2) == [work[1]]
    leaf = list_experiments(os.path.join(ROOT, "Mix"), "geometry_editor")
    assert len(leaf) == 2 and all(t == "geometry_editor" for _, t in leaf)
    with pytest.raises(ValueError):
        list_experiments(os.path.join(ROOT, "Mix"))


def test_compose_transform_matches_reference():
    from geodiffuser_amd.ui_utils import compose_transform
    g = load("G15_exp_folder")
    for i, kw in enumerate(cases.TRANSFORM_CASES):
        t = compose_transform(**kw)
        assert t.dtype == torch.float32 and np.array_equal(t.numpy(), g[f"transform_{i}"]), kw


@pytest.mark.parametrize("etype", ["geometry_editor", "geometry_remover"])
def test_perform_exp_passes_the_reference_configuration(monkeypatch, etype):
    from geodiffuser_amd import editor, large_scale_editor as L
    with open(os.path.join(GOLDEN_DIR, "G16_batch_config.json")) as fh:
        want = json.load(fh)[etype]
    e = cases.exp_case("Mix", 1)
    exp_dict = {"input_image_png": e["image"], "input_mask_png": np.repeat((e["mask"] * 255).astype(np.uint8)[..., None], 3, -1),
                "depth_npy": e["depth"], "transform_npy": e["transform"]}
    seen = {}

    def fake(image, depth, image_mask, transform_in, prompt, **kw):
        seen.update(kw)
        seen["_mask_sum"] = float(image_mask.sum()); seen["_mask_shape"] = list(image_mask.shape)
        seen["_transform_dtype"] = str(transform_in.dtype)
        return [image, image], {}

    monkeypatch.setattr(editor, "perform_geometric_edit", fake)
    images, loss, store = L.perform_exp(exp_dict, edit_type=etype)
    got = {k: v for k, v in seen.items() if k not in ("ldm_stable_model", "tokenizer_model", "scheduler_in")}
    assert json.loads(json.dumps(got, sort_keys=True)) == want
    assert store is None and len(images) == 2
    with pytest.raises(NameError):
        L.edit_config("geometry_stitch")


def test_save_results_writes_the_reference_file_set(tmp_path):
    from geodiffuser_amd import large_scale_editor as L
    from geodiffuser_amd.ui_utils import read_exp, read_image, save_exp
    e = cases.exp_case("Removal", 1)
    folder = save_exp(str(tmp_path), e["image"], e["depth"], e["depth_vis"], e["mask"], e["transform"], h=96, w=64, exp_transform_type="Removal")
    exp_dict = read_exp(folder)
    out = np.clip(e["image"].astype(np.float64) * 0.9 + 3.2, 0, 255)                # float64 like the histogram-matched result
    loss = {0: {"self": {"sim": 1.0, "removal": -0.5}, "cross": {"sim": 0.25}, "num_layers": 20}, 2: {"self": {}, "cross": {}, "num_layers": 20}}
    L.save_results(exp_dict, out, loss, "geometry_remover")
    names = set(os.listdir(folder))
    assert {"loss.log", "loss.pkl", "result_ls.png", "resized_result_ls.png", "experiment.png", "resized_input_image_png.png",
            "resized_input_mask_png.png", "resized_transformed_image_png.png"} <= names
    assert L.load_dictionary(os.path.join(folder, "loss.pkl"))[2]["optimization_step"] == 2
    assert "optimization_step: 2" in open(os.path.join(folder, "loss.log")).read()
    assert np.array_equal(read_image(os.path.join(folder, "result_ls.png")), out.astype(np.uint8))
    # aspect ratio [h, w] = [96, 64] < 1: height stretched by 1/ratio (image_processing.py:100-113)
    assert read_image(os.path.join(folder, "resized_result_ls.png")).shape == (72, 64, 3)
    assert read_image(os.path.join(folder, "experiment.png")).shape == (72, 3 * 64 + 2 * 20, 3)


_BATCH_WORKER = r'''
import os, sys, types
import numpy as np, torch
sys.path.insert(0, %(root)r)
from geodiffuser_amd import diffusion, editor, large_scale_editor as L

def fake_load_model(device="cpu", dtype=torch.float32, **kw):
    torch.manual_seed(100 + int(os.environ["RANK"]))            # every rank starts from DIFFERENT weights: only the broadcast aligns them
    pipe = types.SimpleNamespace(unet=torch.nn.Linear(8, 8), vae=torch.nn.Linear(4, 4), text_encoder=torch.nn.Linear(2, 2))
    return pipe, None, None

def fake_edit(image, depth, image_mask, transform_in, prompt="", ldm_stable_model=None, edit_type="geometry_editor", **kw):
    # a deterministic function of the experiment's inputs AND the model weights (so the result files prove which weights were used)
    w = float(ldm_stable_model.unet.weight.double().sum()) + float(ldm_stable_model.vae.weight.double().sum())
    out = np.clip(image.astype(np.float64) * 0.5 + 40.0 * (image_mask[..., None] > 0.5) + (w %% 7.0), 0, 255)
    log = {0: {"self": {"sim": w}, "cross": {"sim": float(np.asarray(transform_in).sum())}, "num_layers": 16}}
    return [image, out], log

diffusion.load_model = fake_load_model
editor.perform_geometric_edit = fake_edit
L.main(["--root", %(data)r])
print("rank", os.environ["RANK"], "done")
'''


def test_two_rank_gloo_batch_driver_on_experiment_folders(tmp_path):
    """BASELINE configs[2] without the hardware: ``large_scale_editor.main`` as two gloo ranks over a dataset root (category folders ->
    edit types, Rotation_2D skipped), the edit itself stubbed (it needs the GPU).  Every experiment is processed exactly once, by rank
    j mod 2, with RANK 0's weights (the start-up broadcast), and the result files equal those of a single-process run."""
    import shutil
    import subprocess
    import sys
    from geodiffuser_amd import large_scale_editor as L
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for world in (1, 2):
        data = tmp_path / f"w{world}"
        shutil.copytree(ROOT, data)
        script = tmp_path / f"worker{world}.py"
        script.write_text(_BATCH_WORKER % dict(root=repo, data=str(data)))
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29640 + world), WORLD_SIZE=str(world))
        procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True) for r in range(world)]
        logs = [p.communicate(timeout=300)[0] for p in procs]
        for r, (p, o) in enumerate(zip(procs, logs)):
            assert p.returncode == 0, o
            assert f"rank {r} done" in o
        outs[world] = (data, logs)
    data2, logs2 = outs[2]
    work = [os.path.relpath(f, data2).rstrip("/") for f, _ in L.list_experiments(str(data2))]
    assert work == ["Mix/1", "Mix/2", "Removal/1"]
    for j, rel in enumerate(work):                              # edit j ran on rank j mod 2 only
        for r in range(2):
            assert (f"Completed: {data2}/{rel}/" in logs2[r]) == (j % 2 == r), (rel, r)
    assert not os.path.exists(data2 / "Rotation_2D" / "1" / "result_ls.png")          # skipped category (large_scale_editor.py:378-381)
    for rel in work:
        for name in ("result_ls.png", "resized_result_ls.png", "experiment.png", "loss.pkl", "loss.log"):
            a, b = outs[1][0] / rel / name, data2 / rel / name
            assert a.exists() and b.exists(), (rel, name)
            assert a.read_bytes() == b.read_bytes(), (rel, name)      # rank 1's results were computed with rank 0's weights


def test_batched_driver_groups_by_edit_type_and_writes_the_same_result_files(tmp_path, monkeypatch):
    """``large_scale_editor --edits-per-pass 2``: a rank's folders go through geodiffuser_amd.batch.perform_geometric_edit_batch in groups of
    one edit type (editors and removers never share a batch); every folder gets the result files of the one-at-a-time driver.  The edit
    itself is stubbed (it needs the GPU): a deterministic function of the experiment's inputs."""
    import shutil
    import types
    import geodiffuser_amd.batch as GB
    from geodiffuser_amd import diffusion, editor, large_scale_editor as L

    def fake_edit(image, depth, image_mask, transform_in, prompt="", edit_type="geometry_editor", **kw):
        out = np.clip(image.astype(np.float64) * 0.5 + 40.0 * (np.asarray(image_mask)[..., None] > 0.5) + (3.0 if edit_type == "geometry_remover" else 0.0), 0, 255)
        log = {0: {"self": {"sim": float(kw["guidance_scale"])}, "cross": {"sim": float(np.asarray(transform_in).sum())}, "num_layers": 16}}
        return [image, out], log

    seen = []

    def fake_batch(edits, edit_type="geometry_editor", **kw):
        seen.append((edit_type, len(edits)))
        kw.pop("return_loss_log_dict", None)
        return [fake_edit(e["image"], e["depth"], e["image_mask"], e["transform_in"], edit_type=edit_type, **kw) for e in edits]

    monkeypatch.setattr(diffusion, "load_model", lambda **kw: (types.SimpleNamespace(unet=torch.nn.Linear(2, 2), vae=torch.nn.Linear(2, 2),
                                                                                       text_encoder=torch.nn.Linear(2, 2)), None, None))
    monkeypatch.setattr(editor, "perform_geometric_edit", fake_edit)
    monkeypatch.setattr(GB, "perform_geometric_edit_batch", fake_batch)
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(v, raising=False)
    runs = {}
    for per_pass in (1, 2):
        data = tmp_path / f"b{per_pass}"
        shutil.copytree(ROOT, data)
        L.main(["--root", str(data)] + (["--edits-per-pass", "2"] if per_pass > 1 else []))
        runs[per_pass] = data
    assert seen == [("geometry_editor", 2)]                    # Mix/1 + Mix/2 as one batch; the single Removal folder ran on its own
    assert L.group_for_batches([("a", "geometry_editor"), ("r", "geometry_remover"), ("b", "geometry_editor"), ("c", "geometry_editor")], 2) == \
        [(["a", "b"], "geometry_editor"), (["c"], "geometry_editor"), (["r"], "geometry_remover")]
    for rel in ("Mix/1", "Mix/2", "Removal/1"):
        for name in ("result_ls.png", "resized_result_ls.png", "experiment.png", "loss.pkl", "loss.log"):
            a, b = runs[1] / rel / name, runs[2] / rel / name
            assert a.exists() and b.exists() and a.read_bytes() == b.read_bytes(), (rel, name)


def test_folder_io_writes_behind_the_same_bytes_and_raises_worker_errors(tmp_path, monkeypatch):
    """``--io-threads``: reading the next folders ahead and writing results behind on worker threads (FolderIO) leaves byte-identical
    files to the in-line driver (--io-threads 0 = the reference's order of operations), keeps at most max_pending writes in flight, and
    an exception in a worker surfaces from the driver instead of being lost."""
    import shutil
    import threading
    import types
    from geodiffuser_amd import diffusion, editor, large_scale_editor as L

    edit_threads = set()

    def fake_edit(image, depth, image_mask, transform_in, prompt="", edit_type="geometry_editor", **kw):
        edit_threads.add(threading.current_thread().name)
        out = np.clip(image.astype(np.float64) * 0.5 + 40.0 * (np.asarray(image_mask)[..., None] > 0.5), 0, 255)
        return [image, out], {0: {"self": {"sim": 1.0}, "cross": {"sim": float(np.asarray(transform_in).sum())}, "num_layers": 16}}

    monkeypatch.setattr(diffusion, "load_model", lambda **kw: (types.SimpleNamespace(unet=torch.nn.Linear(2, 2), vae=torch.nn.Linear(2, 2),
                                                                                       text_encoder=torch.nn.Linear(2, 2)), None, None))
    monkeypatch.setattr(editor, "perform_geometric_edit", fake_edit)
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(v, raising=False)
    runs = {}
    for threads in (0, 3):
        data = tmp_path / f"io{threads}"
        shutil.copytree(ROOT, data)
        L.main(["--root", str(data), "--io-threads", str(threads)])
        runs[threads] = data
    assert edit_threads == {threading.current_thread().name}             # the edits themselves stay on the caller's thread
    for rel in ("Mix/1", "Mix/2", "Removal/1"):
        names = sorted(os.listdir(runs[0] / rel))
        assert names == sorted(os.listdir(runs[3] / rel))
        for name in names:
            assert (runs[0] / rel / name).read_bytes() == (runs[3] / rel / name).read_bytes(), (rel, name)

    # back-pressure: never more than max_pending unfinished writes
    gate, peak, live = threading.Event(), [0], [0]
    lock = threading.Lock()

    def slow_save(*a, **k):
        with lock:
            live[0] += 1
            peak[0] = max(peak[0], live[0])
        gate.wait(5)
        with lock:
            live[0] -= 1

    monkeypatch.setattr(L, "save_results", slow_save)
    io = L.FolderIO(threads=8, max_pending=2)
    threading.Timer(0.3, gate.set).start()
    for i in range(6):
        io.save({"path_name": str(tmp_path) + "/"}, None, {}, "geometry_editor")
        assert len(io.pending) <= 2
    io.close()
    assert peak[0] <= 2 and io.pending == []

    # a failing writer is reported
    def bad_save(*a, **k):
        raise OSError("disk full")

    monkeypatch.setattr(L, "save_results", bad_save)
    data = tmp_path / "bad"
    shutil.copytree(ROOT, data)
    with pytest.raises(OSError, match="disk full"):
        L.main(["--root", str(data), "--io-threads", "2"])


def test_batched_folder_group_of_mixed_image_sizes_runs_one_batch_per_size(tmp_path, monkeypatch):
    """A batch shares one image size (the UNet pass is one tensor): a group with folders of another size runs them as a batch of their own,
    results and files in folder order."""
    import geodiffuser_amd.batch as GB
    from geodiffuser_amd import large_scale_editor as L
    seen = []

    def fake_batch(edits, edit_type="geometry_editor", **kw):
        sizes = {np.asarray(e["image"]).shape for e in edits}
        assert len(sizes) == 1
        seen.append((sizes.pop()[0], len(edits)))
        return [([e["image"], np.clip(e["image"].astype(np.float64) + 1.0, 0, 255)], {0: {"self": {"sim": 1.0}, "cross": {"sim": 2.0}, "num_layers": 16}}) for e in edits]

    monkeypatch.setattr(GB, "perform_geometric_edit_batch", fake_batch)
    dicts = []
    for j, size in enumerate((64, 32, 64)):
        d = tmp_path / str(j)
        d.mkdir()
        img = np.full((size, size, 3), 10 * (j + 1), dtype=np.uint8)
        dicts.append({"input_image_png": img, "input_mask_png": np.zeros((size, size, 3), dtype=np.uint8), "depth_npy": np.ones((size, size), np.float32),
                      "transform_npy": np.eye(4, dtype=np.float32), "image_shape_npy": np.array([size, size]), "path_name": str(d) + "/"})
    out = L.run_exp_on_folders_batched([str(tmp_path / str(j)) for j in range(3)], "geometry_editor", None, None, None, exp_dicts=dicts)
    assert sorted(seen) == [(32, 1), (64, 2)]
    assert [o[0].shape[0] for o in out] == [64, 32, 64] and [int(o[0][0, 0, 0]) for o in out] == [10, 20, 30]
    for j in range(3):
        assert (tmp_path / str(j) / "result_ls.png").exists() and (tmp_path / str(j) / "loss.pkl").exists()

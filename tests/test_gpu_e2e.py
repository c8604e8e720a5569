"""File-level drivers on the GPU: views/ and sampled/ directories in, the
reference's .npy + CSV outputs out, checked against the CPU oracle."""
from argparse import Namespace

import numpy as np
import pandas as pd
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu


def write_views(root, hu_panels, n_samples=3, views=3):
    from PIL import Image
    ids = []
    for s in range(n_samples):
        sid = f"tree_{s:02d}"
        d = root / sid
        d.mkdir(parents=True)
        for v in range(views):
            Image.fromarray(hu_panels["raw_views"][s, v]).save(d / f"view_{v + 1:02d}.png")
        ids.append(sid)
    return ids


def test_hu_directory_run(gpu, tmp_path, oracle, c_oracle, hu_panels):
    hu = pkg("extract_2d_hu")
    views, out, rep = tmp_path / "views", tmp_path / "2d_hu", tmp_path / "reports"
    ids = write_views(views, hu_panels)
    (views / "tree_01" / "view_09.png").write_bytes(b"not a png")           # unreadable view: logged and skipped
    (views / "broken").mkdir()
    (views / "broken" / "view_01.png").write_bytes(b"junk")                 # sample without usable views
    rc = hu.main(["--views_dir", str(views), "--out_dir", str(out), "--reports_dir", str(rep), "--preprocess_crop",
                  "--use_silhouette", "--expected_views", "3", "--overwrite"])
    assert rc == 0
    df = pd.read_csv(out / "hu_features_table.csv").set_index("id")
    assert list(df.columns) == hu.HU_COLS[1:]
    assert set(df.index) == set(ids) | {"broken"}
    assert df.loc["broken", "note"].startswith("failed") and np.isnan(df.loc["broken", "hu1_mean"])
    assert df.loc["tree_00", "note"] == "ok" and df.loc["tree_01", "note"] == "ok"  # 4 files >= expected 3
    log = (rep / "read_errors.log").read_text()
    assert "view_09.png" in log and "broken" in log
    for s, sid in enumerate(ids):
        masks = [oracle.preprocess_image_cv(hu_panels["raw_views"][s, v], 224, 0.05, "white", True)[1] for v in range(3)]
        rows = c_oracle.hu_batch(np.stack(masks))["hulog7"]
        want = oracle.hu_groupstat(rows)[0]
        got = df.loc[sid, hu.HU_COLS[1:15]].to_numpy(dtype=np.float64).astype(np.float32)
        # per-view rows may differ from the oracle's by one float32 ulp (log10), the
        # float32 mean/std of 3 rows then by a few ulps
        np.testing.assert_allclose(got[:7], want[:7], rtol=1e-6)
        np.testing.assert_allclose(got[7:], want[7:], rtol=1e-4, atol=1e-6)
        saved = np.load(out / f"{sid}_hu.npy")
        assert saved.dtype == np.float32 and saved.shape == (14,)
        np.testing.assert_array_equal(saved, got)
    assert (rep / "hu_feature_stats.csv").exists() and (out / "extract_2d_hu_config.yaml").exists()


def test_hu_worker_dict_contract(gpu, tmp_path, hu_panels):
    hu = pkg("extract_2d_hu")
    views = tmp_path / "views"
    write_views(views, hu_panels, n_samples=1, views=2)
    cfg = {"out_dir": str(tmp_path / "o"), "resolution": 224, "use_silhouette": False, "preprocess_crop": False,
           "crop_margin": 0.05, "bg_color": "white", "expected_views": 12, "overwrite": True,
           "reports_dir": str(tmp_path / "r")}
    res = hu.process_sample_worker(("tree_00", str(views / "tree_00"), cfg))
    assert set(res) == {"success", "id", "vec", "note"}
    assert res["success"] and res["id"] == "tree_00" and len(res["vec"]) == 14 and res["note"] == "partial_views"
    miss = hu.process_sample_worker(("nope", str(views / "nope"), cfg))
    assert not miss["success"] and miss["vec"] is None and miss["note"].startswith("failed: No view images")


def test_pca_directory_run_and_merge(gpu, tmp_path, oracle):
    pca, mf, synth = pkg("extract_3d_pca_features"), pkg("merge_features"), pkg("synth")
    sampled, out, rep = tmp_path / "processed" / "sampled", tmp_path / "desc", tmp_path / "reports"
    sampled.mkdir(parents=True)
    clouds = synth.points_host(5, 1024, seed=8)
    for k in range(5):
        np.save(sampled / f"cloud_{k}.npy", clouds[k])
    np.save(sampled / "cloud_small.npy", np.zeros((2, 3), np.float32))        # dropped from the table
    np.save(sampled / "cloud_big.npy", synth.points_host(1, 5000, seed=9)[0])  # other size: ragged batch
    pd.DataFrame({"id": ["cloud_0"], "raw_path": ["/raw/cloud_0.pts"]}).to_csv(tmp_path / "processed" / "manifest.csv", index=False)
    rc = pca.main(["--sampled_dir", str(sampled), "--out_dir", str(out), "--reports_dir", str(rep), "--n_bins", "8"])
    assert rc == 0
    df = pd.read_csv(out / "descriptors_table_pca.csv").set_index("id")
    assert list(df.columns) == pca.column_names(8)[1:]
    assert "cloud_small" not in df.index and len(df) == 6
    assert df.loc["cloud_0", "raw_path"] == "/raw/cloud_0.pts" and df.loc["cloud_big", "n_points"] == 5000
    for k in range(5):
        want = oracle.compute_pca_features(clouds[k].astype(np.float64))
        got = df.loc[f"cloud_{k}", pca.column_names(8)[3:]].to_numpy(dtype=np.float64)
        np.testing.assert_allclose(got[:9], want[:9], rtol=1e-6)
        assert np.abs(got[9:] - want[9:]).max() * 1024 <= 2.5
        np.testing.assert_array_equal(np.load(out / f"pca_cloud_{k}.npy"), got.astype(np.float32))
    assert "cloud_small" in (rep / "read_errors.log").read_text()
    # merge with a Hu table: the reference's joined layout
    hu_cols = pkg("extract_2d_hu").HU_COLS
    pd.DataFrame([[f"cloud_{k}"] + [float(k)] * 14 + ["ok"] for k in range(3)], columns=hu_cols).to_csv(tmp_path / "hu.csv", index=False)
    mf.main(Namespace(hu_csv=str(tmp_path / "hu.csv"), pca_csv=str(out / "descriptors_table_pca.csv"), out=str(tmp_path / "features.csv")))
    m = pd.read_csv(tmp_path / "features.csv").set_index("id")
    assert len(m) == 6 and m.loc["cloud_4", "hu1_mean"] == 0 and m.loc["cloud_2", "hu1_mean"] == 2.0
    assert m.loc["cloud_1", "lambda1"] == pytest.approx(df.loc["cloud_1", "lambda1"])


def test_hu_directory_run_mixed_files(gpu, tmp_path, oracle, c_oracle, hu_panels):
    """One sample whose views come as RGB PNG, RGBA PNG, palette PNG, JPEG and a
    PNG of another size: native decoder and PIL side by side, rows in file-name
    order, every view resized to the target like the reference does."""
    from PIL import Image
    hu = pkg("extract_2d_hu")
    views, out, rep = tmp_path / "views", tmp_path / "2d_hu", tmp_path / "reports"
    raw = hu_panels["raw_views"]
    d = views / "tree_mixed"
    d.mkdir(parents=True)
    Image.fromarray(raw[0, 0]).convert("RGB").save(d / "view_01.png")
    Image.fromarray(raw[0, 1]).convert("RGBA").save(d / "view_02.png")
    Image.fromarray(raw[0, 2]).convert("P").save(d / "view_03.png")
    Image.fromarray(raw[1, 0]).save(d / "view_04.jpg", quality=95)
    Image.fromarray(raw[1, 1][:180, :200]).save(d / "view_05.png")
    Image.fromarray(raw[1, 2]).save(d / "view_06.png")
    d2 = views / "tree_plain"
    d2.mkdir()
    for v in range(3):
        Image.fromarray(raw[2, v]).save(d2 / f"view_{v + 1:02d}.png")
    for silhouette in (True, False):
        argv = ["--views_dir", str(views), "--out_dir", str(out), "--reports_dir", str(rep), "--preprocess_crop",
                "--expected_views", "3", "--overwrite"] + (["--use_silhouette"] if silhouette else [])
        assert hu.main(argv) == 0
        df = pd.read_csv(out / "hu_features_table.csv").set_index("id")
        assert df.loc["tree_mixed", "note"] == "ok" and df.loc["tree_plain", "note"] == "ok"
        for sid, folder in (("tree_mixed", d), ("tree_plain", d2)):
            rows = []
            for f in sorted(folder.iterdir()):
                img = np.asarray(Image.open(f).convert("L"))
                pre, mask = oracle.preprocess_image_cv(img, 224, 0.05, "white", True)
                rows.append(c_oracle.hu_batch((mask if silhouette else pre)[None])["hulog7"][0])
            want = oracle.hu_groupstat(np.stack(rows))[0]
            got = df.loc[sid, hu.HU_COLS[1:15]].to_numpy(dtype=np.float64).astype(np.float32)
            np.testing.assert_allclose(got[:7], want[:7], rtol=2e-6)
            np.testing.assert_allclose(got[7:], want[7:], rtol=1e-4, atol=1e-6)


def test_png_inflated_on_the_device_equals_the_host_decode(gpu, tmp_path, hu_panels, monkeypatch):
    """The directory run with the PNG streams inflated on the device (the default) and with FX_PNG_DEVICE=0 (native
    host threads) writes the same table; a PNG whose pixel stream is damaged behind an intact container is declined
    by the device, fails in PIL too, is logged and left out in both runs."""
    import struct
    import zlib
    from PIL import Image
    hu = pkg("extract_2d_hu")
    raw = hu_panels["raw_views"]
    views = tmp_path / "views"
    for s in range(6):
        d = views / f"tree_{s:02d}"
        d.mkdir(parents=True)
        for v in range(4):
            img = Image.fromarray(raw[s % raw.shape[0], v % raw.shape[1]])
            (img if (s + v) % 3 else img.convert("RGB")).save(d / f"view_{v + 1:02d}.png", compress_level=(s + v) % 10)
    # damage one file's zlib stream (last byte of its Adler-32) and repair the chunk's CRC
    victim = views / "tree_03" / "view_02.png"
    data = bytearray(victim.read_bytes())
    at = data.rindex(b"IDAT")
    n = struct.unpack(">I", data[at - 4:at])[0]
    data[at + 4 + n - 1] ^= 0x55
    data[at + 4 + n:at + 8 + n] = struct.pack(">I", zlib.crc32(bytes(data[at:at + 4 + n])) & 0xFFFFFFFF)
    victim.write_bytes(bytes(data))
    tables = []
    for tag, flag in (("dev", "1"), ("host", "0")):
        monkeypatch.setenv("FX_PNG_DEVICE", flag)
        out, rep = tmp_path / f"out_{tag}", tmp_path / f"rep_{tag}"
        assert hu.main(["--views_dir", str(views), "--out_dir", str(out), "--reports_dir", str(rep), "--preprocess_crop",
                        "--use_silhouette", "--expected_views", "4", "--overwrite"]) == 0
        tables.append(pd.read_csv(out / "hu_features_table.csv").set_index("id"))
        assert "view_02.png" in (rep / "read_errors.log").read_text()
    a, b = tables
    assert list(a.index) == list(b.index) and len(a) == 6
    np.testing.assert_array_equal(a[hu.HU_COLS[1:15]].to_numpy(), b[hu.HU_COLS[1:15]].to_numpy())
    assert list(a["note"]) == list(b["note"])


def test_allgather_features_on_a_one_rank_communicator(gpu):
    """fx_allgather_features with an RCCL communicator made by the caller (one
    rank here; the N-rank case is the same call)."""
    import ctypes
    import torch
    lib = pkg("_lib")
    pkg("engine").synth_points(1, 8, device=gpu)            # library initialised on the device
    rccl = None
    with open("/proc/self/maps") as fh:                      # the copy PyTorch has already loaded
        for line in fh:
            if "librccl" in line:
                rccl = ctypes.CDLL(line.split()[-1])
                break
    if rccl is None:
        rccl = ctypes.CDLL("librccl.so.1")

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        local = torch.arange(5 * 24, dtype=torch.float32, device=gpu).reshape(5, 24) / 7
        table = torch.full((5, 24), -1.0, dtype=torch.float32, device=gpu)
        stream = torch.cuda.current_stream().cuda_stream
        lib.call("fx_allgather_features", local.data_ptr(), 5, 24, table.data_ptr(), comm, stream)
        torch.cuda.synchronize()
        assert torch.equal(table, local)
        with pytest.raises(lib.FxError, match="rccl_comm is NULL"):
            lib.call("fx_allgather_features", local.data_ptr(), 5, 24, table.data_ptr(), None, stream)
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_exchange_helpers_on_a_one_rank_rccl_group(gpu, monkeypatch):
    """The collectives of the N-rank run -- the feature all-gather (blocking, and two deep through GatherPipeline), the
    id-tagged gather of units assigned by cost, the limb all-reduce of row bands and the description of the ranks --
    issued on torch.distributed's RCCL backend with a group of one rank: same calls, same tensors, one participant.
    (What several ranks add is covered over gloo on the CPU.)"""
    import socket
    import torch
    import torch.distributed as td
    fxdist = pkg("dist")
    eng = pkg("engine")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    assert not td.is_initialized()
    td.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=gpu)
    try:
        monkeypatch.setattr(fxdist, "MIN_RANKS_FOR_COLLECTIVES", 1)
        imgs = eng.synth_shapes(37, 64, 64, seed=5, device=gpu)
        feats = eng.hu_batch(imgs, thresh=127, on_value=255)["hulog7"]
        table = fxdist.all_gather_rows(feats, 37)
        assert table.data_ptr() != feats.data_ptr() and torch.equal(table, feats)
        pipe = fxdist.GatherPipeline(depth=2)
        assert pipe.active
        blocks = []
        for k in range(5):
            blocks.append(eng.hu_batch(imgs[k:k + 20], thresh=127, on_value=255)["hulog7"])
            pipe.submit(blocks[-1])
        assert torch.equal(pipe.drain(), blocks[-1])
        rows = torch.arange(6 * 17, dtype=torch.float32, device=gpu).reshape(6, 17)
        order = [4, 0, 5, 2, 1, 3]
        tagged = fxdist.all_gather_tagged_rows(rows, order, 8, 6)
        assert torch.equal(tagged[order], rows)
        with pytest.raises(ValueError, match="missing"):
            fxdist.all_gather_tagged_rows(rows[:5], order[:5], 8, 6)
        tall = eng.synth_shapes(2, 1024, 512, seed=9, device=gpu)
        whole = eng.hu_batch(tall, thresh=127, on_value=255)["hulog7"]
        banded = fxdist.hu_rowbands(tall, 0, 127, 255)["hulog7"]
        assert torch.equal(banded, whole)
        report = fxdist.describe_ranks(1.25, torch.cuda.get_device_name(gpu))
        assert report["ranks"] == 1 and report["backend"] == "nccl" and len(report["devices"]) == 1
        torch.cuda.synchronize()
    finally:
        td.destroy_process_group()


def test_clouds_to_hu_table_on_the_device(gpu, oracle, c_oracle, tmp_path):
    """Clouds -> rendered views -> preprocessing -> Hu -> per-sample statistics
    without leaving HBM equals the chain through PNG files: the oracle's
    renderer, preprocess_image and Hu per view, float32 mean / std per sample."""
    hu, rv = pkg("extract_2d_hu"), pkg("render_views")
    rng = np.random.default_rng(21)
    clouds = [(rng.normal(size=(1024, 3)) * s).astype(np.float32) for s in ([0.3, 0.25, 0.6], [0.5, 0.1, 0.2])]
    views, res = 6, 128
    for use_sil, crop in ((False, False), (True, True)):
        got, rendered = hu.hu_from_clouds(clouds, views, res, use_silhouette=use_sil, preprocess_crop=crop, return_views=True)
        assert got.shape == (2, 14) and got.dtype == np.float32
        for b, c in enumerate(clouds):
            p64 = c.astype(np.float64)
            mins, maxs, cen, _ = oracle.bbox_and_center(p64)
            poses = oracle.compute_camera_poses(views, mins, maxs, cen)
            rows = []
            for v in range(views):
                img = oracle.render_view(p64, poses[v], res, rv.marker_stamp(2.0))
                np.testing.assert_array_equal(rendered[b, v].cpu().numpy(), img)
                pre, mask = oracle.preprocess_image_cv(img, res, 0.05, "white", crop)
                rows.append(c_oracle.hu_batch((mask if use_sil else pre)[None], thresh=-1)["hulog7"][0])
            rows = np.stack(rows).astype(np.float32)
            want = np.concatenate([rows.mean(axis=0), rows.std(axis=0)]).astype(np.float32)
            np.testing.assert_allclose(got[b], want, rtol=1e-5, atol=1e-6, err_msg=f"cloud {b} silhouette {use_sil}")
    # the same numbers through the files: script-rendered PNGs read by the Hu script's worker
    sampled = tmp_path / "sampled"
    sampled.mkdir()
    np.save(sampled / "tree_a.npy", clouds[0])
    assert rv.main(["--sampled_dir", str(sampled), "--out_dir", str(tmp_path / "views"), "--views", str(views),
                    "--resolution", str(res), "--reports_dir", str(tmp_path / "rep"), "--n_preview", "0"]) == 0
    cfg = {"out_dir": str(tmp_path / "o"), "resolution": res, "use_silhouette": False, "preprocess_crop": False,
           "crop_margin": 0.05, "bg_color": "white", "expected_views": views, "overwrite": True,
           "reports_dir": str(tmp_path / "rep")}
    rec = hu.process_sample_worker(("tree_a", str(tmp_path / "views" / "tree_a"), cfg))
    direct = hu.hu_from_clouds(clouds[:1], views, res)
    np.testing.assert_array_equal(np.asarray(rec["vec"], dtype=np.float32), direct[0])


def test_preview_collages_equal_the_reference_held_ones(gpu, tmp_path, hu_panels, capsys):
    """main() with the production flags and --n_preview reproduces, pixel for pixel, the 21 preview collages
    the reference's run left under reports/sample_images (1120 x 224: views 01, 05, 09, the cropped gray view,
    its silhouette; all gray, R = G = B), from the three raw views each of them shows."""
    from PIL import Image
    hu = pkg("extract_2d_hu")
    views, out, rep = tmp_path / "views", tmp_path / "2d_hu", tmp_path / "reports"
    ids = hu_panels["ids"]
    for s, sid in enumerate(ids):
        d = views / sid
        d.mkdir(parents=True)
        for v, name in enumerate(("view_01.png", "view_05.png", "view_09.png")):
            Image.fromarray(hu_panels["raw_views"][s, v]).save(d / name)
    rc = hu.main(["--views_dir", str(views), "--out_dir", str(out), "--reports_dir", str(rep), "--preprocess_crop",
                  "--use_silhouette", "--expected_views", "3", "--overwrite", "--n_preview", str(len(ids) + 5)])
    assert rc == 0
    for s, sid in enumerate(ids):
        got = np.asarray(Image.open(rep / "sample_images" / f"{sid}_hu_preview.png"))
        want = np.concatenate([hu_panels["raw_views"][s, 0], hu_panels["raw_views"][s, 1], hu_panels["raw_views"][s, 2],
                               hu_panels["preproc"][s], hu_panels["masks"][s]], axis=1)
        assert got.shape == (224, 1120, 3) and got.dtype == np.uint8
        np.testing.assert_array_equal(got, np.repeat(want[:, :, None], 3, axis=2), err_msg=sid)
    assert capsys.readouterr().out.count("[mask-clean]") == 5
    # without --use_silhouette the last panel is the preprocessed image, binarised at 128 like every last panel
    rc = hu.main(["--views_dir", str(views), "--out_dir", str(out), "--reports_dir", str(tmp_path / "rep2"),
                  "--preprocess_crop", "--expected_views", "3", "--overwrite", "--n_preview", "2", "--max_samples", "2"])
    assert rc == 0
    first = sorted(ids)[0]
    got = np.asarray(Image.open(tmp_path / "rep2" / "sample_images" / f"{first}_hu_preview.png"))
    np.testing.assert_array_equal(got[:, 896:], np.where(got[:, 672:896] > 128, 255, 0).astype(np.uint8))

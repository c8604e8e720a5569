"""View renderer on the device against views drawn by the reference's own
render_fallback_matplotlib (golden: matplotlib's Agg output) and against the
oracle's restatement on seeded clouds: identical pixels."""
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import pkg
from test_oracle import render_cases

pytestmark = pytest.mark.gpu


def test_golden_views_of_the_reference(gpu):
    rv = pkg("render_views")
    done = {}
    for k, pts, poses, v, res, ps, bg, views, img, stamps in render_cases():
        key = (pts.tobytes(), res, ps, bg, views)
        if key not in done:
            done[key] = rv.render_views_batch([pts], views, res, ps, bg).cpu().numpy()[0]
        got = done[key][v]
        diff = int((got != img).sum())
        assert diff == 0, f"case {k}: {diff} pixels differ ({len(pts)} points, {res} px, size {ps}, {bg}, view {v})"
        # the single-view function with the reference's pose dictionary
        pose = {"position": poses[v, 0:3], "lookat": poses[v, 3:6], "up": poses[v, 6:9]}
        one = rv.render_fallback_matplotlib(pts.astype(np.float64), pose, res, ps, bg, False)
        assert one.mode == ("RGB" if bg == "white" else "RGBA") and one.size == (res, res)
        np.testing.assert_array_equal(np.asarray(one.convert("RGB"))[:, :, 0], img)


def test_batches_against_oracle(gpu, oracle):
    """Ragged batches, clouds larger than one LDS chunk of snapped positions,
    degenerate clouds (a single point, all points equal, a line)."""
    rv = pkg("render_views")
    rng = np.random.default_rng(77)
    clouds = []
    for n in (1, 2, 50, 1024, 1024, 3000, 4097):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        clouds.append(((rng.normal(size=(n, 3)) * np.exp(rng.uniform(-2, 1, size=3))) @ q.T + rng.normal(size=3) * 3).astype(np.float32))
    clouds.append(np.tile(np.array([[0.5, -1.0, 2.0]], dtype=np.float32), (30, 1)))
    clouds.append(np.c_[np.linspace(-1, 1, 200), np.zeros(200), np.zeros(200)].astype(np.float32))
    # (0.5: a 1 x 1 stamp, where coinciding points of one step must still blend one after the other;
    # the last two: banded canvas)
    for res, ps, views in ((224, 2.0, 12), (224, 0.5, 3), (96, 5.0, 4), (384, 1.0, 3), (512, 2.0, 2), (700, 4.0, 2)):
        got = rv.render_views_batch(clouds, views, res, ps).cpu().numpy()
        assert got.shape == (len(clouds), views, res, res)
        stamp = rv.marker_stamp(ps)
        for b, c in enumerate(clouds):
            p64 = c.astype(np.float64)
            mins, maxs, cen, _ = oracle.bbox_and_center(p64)
            poses = oracle.compute_camera_poses(views, mins, maxs, cen, 2.0, 42)
            for v in (0, views - 1):
                want = oracle.render_view(p64, poses[v], res, stamp)
                diff = int((got[b, v] != want).sum())
                assert diff == 0, f"cloud {b} ({len(c)} points) view {v} at {res} px, size {ps}: {diff} pixels differ"


def test_bounds_poses_and_errors(gpu, oracle):
    rv, eng = pkg("render_views"), pkg("engine")
    rng = np.random.default_rng(5)
    clouds = [rng.normal(size=(n, 3)).astype(np.float32) * 10 for n in (1, 7, 1000, 20000)]
    pts, off = rv.pack_clouds(clouds, gpu)
    b = eng.cloud_bounds(pts, off).cpu().numpy()
    for k, c in enumerate(clouds):
        mins, maxs, cen, _ = oracle.bbox_and_center(c.astype(np.float64))
        np.testing.assert_array_equal(b[k], np.concatenate([mins, maxs, cen]))
    # the camera table computed on the device is the host's, bit for bit (same float64 operations)
    flat = np.stack([np.array([0.5, -1.0, 2.0] * 3), np.concatenate([[-1.0, 0, 0], [1.0, 0, 0], [0.0, 0, 0]])])
    for views in (1, 5, 12, 64):
        hb = np.concatenate([b, flat])              # + a single point and a line along x
        cam_h, scale_h = rv.camera_table(hb, views, 224, 2.0, 42)
        cam_d, scale_d = eng.camera_table(torch.from_numpy(hb).to(gpu), rv.view_trig(views, 42), 2.0, 224)
        np.testing.assert_array_equal(cam_d.cpu().numpy(), cam_h)
        np.testing.assert_array_equal(scale_d.cpu().numpy(), scale_h)
    mins, maxs, cen, _ = rv.bbox_and_center(clouds[2].astype(np.float64))
    for views, levels in ((12, None), (5, None), (2, None), (7, [10.0, 89.99])):
        mine = rv.compute_camera_poses(views, mins, maxs, cen, 2.0, 42, levels)
        ref = oracle.compute_camera_poses(views, mins, maxs, cen, 2.0, 42, levels)
        assert len(mine) == len(ref) == views
        for a, r in zip(mine, ref):
            for key in ("position", "lookat", "up"):
                np.testing.assert_array_equal(a[key], r[key])
    with pytest.raises(ValueError, match="up to 7 x 7"):
        rv.render_views_batch(clouds[:1], 3, 64, 400.0)
    with pytest.raises(ValueError, match="empty point cloud"):
        rv.render_views_batch([np.zeros((0, 3), np.float32)], 3, 64)
    with pytest.raises(pkg("_lib").FxError):
        rv.render_views_batch(clouds[:1], 3, 5000)


def test_directory_run_and_views_feed_the_hu_path(gpu, tmp_path, oracle):
    """The script writes what the reference's writes (views, manifest, config),
    and the views it keeps on the device give the Hu features of the PNGs."""
    import pandas as pd
    from PIL import Image
    rv, hu = pkg("render_views"), pkg("extract_2d_hu")
    rng = np.random.default_rng(9)
    sampled = tmp_path / "sampled"
    sampled.mkdir()
    clouds = {}
    for name in ("Buche_1", "Eiche_2", "Kiefer_3"):
        c = (rng.normal(size=(1024, 3)) * [0.3, 0.25, 0.6]).astype(np.float32)
        np.save(sampled / f"{name}.npy", c)
        clouds[name] = c
    np.save(sampled / "broken.npy", np.zeros((5, 2), np.float32))
    out = tmp_path / "views"
    rc = rv.main(["--sampled_dir", str(sampled), "--out_dir", str(out), "--views", "6", "--resolution", "128",
                  "--reports_dir", str(tmp_path / "reports"), "--n_preview", "2"])
    assert rc == 0
    man = pd.read_csv(out / "manifest_views.csv")
    assert list(man.columns) == ["id", "sample_path", "views_dir", "n_views", "note", "rendered_timestamp"]
    assert sorted(man["id"]) == ["Buche_1", "Eiche_2", "Kiefer_3", "broken"]
    assert man.set_index("id").loc["broken", "note"].startswith("failed: invalid shape")
    assert (out / "render_config.yaml").exists() and (tmp_path / "reports" / "read_errors.log").exists()
    assert len(list((tmp_path / "reports" / "sample_images").glob("*_preview.png"))) == 2
    batch = rv.render_views_batch([clouds["Eiche_2"]], 6, 128).cpu().numpy()[0]
    for v in range(6):
        png = np.asarray(Image.open(out / "Eiche_2" / f"view_{v + 1:02d}.png").convert("L"))
        np.testing.assert_array_equal(png, batch[v])
    mins, maxs, cen, _ = oracle.bbox_and_center(clouds["Eiche_2"].astype(np.float64))
    pose = oracle.compute_camera_poses(6, mins, maxs, cen)[3]
    np.testing.assert_array_equal(batch[3], oracle.render_view(clouds["Eiche_2"].astype(np.float64), pose, 128, rv.marker_stamp(2.0)))


def test_one_bad_sample_does_not_take_the_batch_with_it(gpu, tmp_path):
    """The reference's worker catches per sample: a cloud the renderer refuses
    (NaN coordinates, a marker too large for the device) or a file that cannot be
    read gets a "failed: ..." record and a line in read_errors.log; every other
    sample of the batch is rendered and gets its manifest row."""
    rv = pkg("render_views")
    rng = np.random.default_rng(5)
    sampled, out, reports = tmp_path / "sampled", tmp_path / "views", tmp_path / "reports"
    sampled.mkdir()
    good = [rng.normal(size=(300, 3)).astype(np.float32) for _ in range(3)]
    np.save(sampled / "a_good.npy", good[0])
    np.save(sampled / "b_empty.npy", np.zeros((0, 3), np.float32))          # fails at load time
    np.save(sampled / "c_good.npy", good[1])
    (sampled / "d_broken.npy").write_bytes(b"not an npy file")              # fails at load time
    np.save(sampled / "e_good.npy", good[2])
    common = (str(out), 96, 4, 2.0, "white", False, False, 2.0, 42, True, str(reports))
    tasks = [(str(p),) + common for p in sorted(sampled.glob("*.npy"))]
    # a task with a marker the device renderer does not take: fails on the device side of the batch
    tasks.append((str(sampled / "a_good.npy"), str(tmp_path / "views2"), 96, 4, 400.0, "white", False, False, 2.0, 42, True,
                  str(reports)))
    res = rv.render_samples(tasks)
    assert [r["success"] for r in res] == [True, False, True, False, True, False]
    assert all(r["record"]["note"] == "ok" for r in res if r["success"])
    assert all(r["record"]["note"].startswith("failed: ") for r in res if not r["success"])
    for r in res:
        if r["success"]:
            assert len(r["saved_paths"]) == 4 and all(Path(p).exists() for p in r["saved_paths"])
    log = (reports / "read_errors.log").read_text().splitlines()
    assert len(log) == 3 and all(len(ln.split("\t")) == 3 for ln in log)
    # the script still writes a manifest row for every sample
    rc = rv.main(["--sampled_dir", str(sampled), "--out_dir", str(out), "--views", "4", "--resolution", "96",
                  "--reports_dir", str(reports), "--overwrite", "--n_preview", "0"])
    assert rc == 0
    import pandas as pd
    man = pd.read_csv(out / "manifest_views.csv")
    assert len(man) == 5 and sorted(man["note"].str.startswith("failed").tolist()) == [False, False, False, True, True]

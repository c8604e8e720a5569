"""PNG decode on the device (fx_png_stage + fx_png_inflate_gray_u8) against PIL's Image.open(path).convert("L")
(reference src/features/extract_2d_hu.py:84-98): byte for byte, or a status that hands the file to the host."""
import struct
import zlib

import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu


def chunk(kind: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)


def write_png(path, pixels: np.ndarray, filters, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, idat_split=0, adler_xor=0):
    """A PNG written by hand: pixels [H, W] or [H, W, C] uint8, one filter type per row (the filtered bytes are
    computed here), the zlib stream optionally cut into several IDAT chunks or given a wrong Adler-32."""
    px = pixels if pixels.ndim == 3 else pixels[:, :, None]
    H, W, C = px.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[C]
    rows = px.reshape(H, W * C).astype(np.int32)
    raw = bytearray()
    for y in range(H):
        ft = filters[y % len(filters)]
        cur = rows[y]
        up = rows[y - 1] if y else np.zeros_like(cur)
        left = np.concatenate([np.zeros(C, dtype=np.int32), cur[:-C]])
        upleft = np.concatenate([np.zeros(C, dtype=np.int32), up[:-C]])
        if ft == 0:
            pred = 0
        elif ft == 1:
            pred = left
        elif ft == 2:
            pred = up
        elif ft == 3:
            pred = (left + up) >> 1
        else:
            p = left + up - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, upleft))
        raw.append(ft)
        raw += ((cur - pred) & 255).astype(np.uint8).tobytes()
    co = zlib.compressobj(level, zlib.DEFLATED, 15, 9, strategy)
    z = bytearray(co.compress(bytes(raw)) + co.flush())
    if adler_xor:
        z[-1] ^= adler_xor
    z = bytes(z)
    parts = [z] if not idat_split else [z[i:i + idat_split] for i in range(0, len(z), idat_split)]
    body = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, ctype, 0, 0, 0))
    body += chunk(b"tEXt", b"Comment\x00by hand")
    for part in parts:
        body += chunk(b"IDAT", part)
    body += chunk(b"IEND", b"")
    path.write_bytes(body)


def view_like(rng, H, W):
    yy, xx = np.mgrid[0:H, 0:W]
    shape = ((xx - W / 2) ** 2 / (W / 5) ** 2 + (yy - H / 2) ** 2 / (H / 3) ** 2) < 1
    return np.clip(235.0 - shape * 195.0 + rng.normal(size=(H, W)) * 8.0, 0, 255).round().astype(np.uint8)


def decode_on_device(paths, H, W, gpu, workers=4):
    ingest, eng = pkg("ingest"), pkg("engine")
    staged = ingest.stage_png_batch([str(p) for p in paths], H, W, workers)
    got = eng.png_inflate_batch(staged, H, W, gpu)
    return staged, got["rows"], got["gray"].cpu().numpy(), got["status"].cpu().numpy()


def test_files_written_by_pil(gpu, tmp_path):
    """What a renderer writes: gray, gray + alpha, RGB, RGBA at every compression level (0 = stored blocks), noisy
    views, flat images (long matches), a tiny image (one fixed-code block)."""
    from PIL import Image
    rng = np.random.default_rng(5)
    H, W = 224, 224
    paths = []
    for k, mode in enumerate(["L", "LA", "RGB", "RGBA"]):
        for level in (0, 1, 6, 9):
            g = view_like(rng, H, W)
            if mode == "L":
                img = Image.fromarray(g)
            else:
                c = {"LA": 2, "RGB": 3, "RGBA": 4}[mode]
                arr = np.stack([np.clip(g.astype(np.int32) + 13 * j, 0, 255).astype(np.uint8) for j in range(c)], axis=2)
                if k % 2:
                    arr[..., :] = rng.integers(0, 256, arr.shape, dtype=np.uint8)
                img = Image.fromarray(arr, mode)
            p = tmp_path / f"{mode}_{level}.png"
            img.save(p, compress_level=level)
            paths.append(p)
    flat = np.full((H, W), 255, dtype=np.uint8)
    flat[60:150, 80:170] = 40
    for level in (1, 9):
        p = tmp_path / f"flat_{level}.png"
        Image.fromarray(flat).save(p, compress_level=level)
        paths.append(p)
    p = tmp_path / "optimized.png"
    Image.fromarray(view_like(rng, H, W)).save(p, optimize=True)
    paths.append(p)
    staged, rows, gray, status = decode_on_device(paths, H, W, gpu)
    assert (staged["status"] == 0).all() and len(rows) == len(paths)
    assert (staged["offset"] % 16 == 0).all()
    assert (status == 0).all(), status
    for i, p in enumerate(paths):
        want = np.asarray(Image.open(p).convert("L"))
        np.testing.assert_array_equal(gray[i], want, err_msg=p.name)


@pytest.mark.parametrize("channels", [1, 2, 3, 4])
def test_every_filter_type_and_deflate_strategy(gpu, tmp_path, channels):
    """Rows filtered with each of the five types (and a mix), streams from every zlib strategy (Z_FIXED: the fixed
    code; Z_HUFFMAN_ONLY: no matches; Z_RLE: distance-one matches), the stream cut into many IDAT chunks."""
    from PIL import Image
    rng = np.random.default_rng(channels)
    H, W = 48, 80
    px = rng.integers(0, 256, (H, W, channels), dtype=np.uint8) if channels > 1 else view_like(rng, H, W)
    if channels > 1:
        px[:, :, 0] = view_like(rng, H, W)
    paths = []
    for name, filters in [("none", [0]), ("sub", [1]), ("up", [2]), ("avg", [3]), ("paeth", [4]), ("mix", [4, 1, 3, 2, 0, 4, 4])]:
        for sname, strat in [("default", zlib.Z_DEFAULT_STRATEGY), ("fixed", zlib.Z_FIXED), ("huff", zlib.Z_HUFFMAN_ONLY),
                             ("rle", zlib.Z_RLE)]:
            p = tmp_path / f"{name}_{sname}.png"
            write_png(p, px, filters, level=6, strategy=strat, idat_split=(0 if sname != "rle" else 100))
            paths.append(p)
    staged, rows, gray, status = decode_on_device(paths, H, W, gpu)
    assert (staged["status"] == 0).all() and (status == 0).all(), (staged["status"], status)
    for i, p in enumerate(paths):
        want = np.asarray(Image.open(p).convert("L"))
        np.testing.assert_array_equal(gray[i], want, err_msg=p.name)


def test_what_the_device_does_not_decode_gets_a_status(gpu, tmp_path):
    """A wrong Adler-32, a stream cut short, a stream for another image size, a filter type above 4: a status, and the
    files around them are decoded as ever.  Files the container check rejects never reach the device."""
    from PIL import Image
    rng = np.random.default_rng(9)
    H, W = 40, 56
    good = view_like(rng, H, W)
    paths = [tmp_path / f"f{k}.png" for k in range(8)]
    write_png(paths[0], good, [4])
    write_png(paths[1], good, [1], adler_xor=0x5A)
    write_png(paths[2], good, [2])
    # the stream of a shorter image under this size's header: the inflate ends early
    write_png(paths[3], good[:20], [0])
    data = paths[3].read_bytes()
    ih = data.index(b"IHDR")
    hdr = struct.pack(">IIBBBBB", W, H, 8, 0, 0, 0, 0)
    data = data[:ih + 4] + hdr + struct.pack(">I", zlib.crc32(b"IHDR" + hdr) & 0xFFFFFFFF) + data[ih + 4 + 13 + 4:]
    paths[3].write_bytes(data)
    write_png(paths[4], good, [7])          # no such filter
    write_png(paths[5], good, [3, 0])
    paths[6].write_bytes(b"not a png at all")
    Image.fromarray(good).convert("P").save(paths[7])   # palette: the container check leaves it to PIL
    staged, rows, gray, status = decode_on_device(paths, H, W, gpu)
    assert list(staged["status"][:6]) == [0] * 6 and staged["status"][6] != 0 and staged["status"][7] != 0
    assert list(rows) == [0, 1, 2, 3, 4, 5]
    assert status[0] == 0 and status[2] == 0 and status[5] == 0
    assert status[1] != 0 and status[3] != 0 and status[4] != 0
    for i in (0, 2, 5):
        np.testing.assert_array_equal(gray[i], good)


def test_entries_a_caller_filled_in_wrongly_get_a_status(gpu, tmp_path):
    """The device entry point takes the three per-file arrays from its caller: a channel count outside 1 .. 4 (or more
    than a row of 4096 bytes holds), a negative length and a length below the six bytes of the smallest zlib stream
    end with a status without a byte of the stream being read; the files around them decode as ever."""
    rng = np.random.default_rng(21)
    H, W = 32, 48
    good = view_like(rng, H, W)
    paths = [tmp_path / f"g{k}.png" for k in range(7)]
    for k, p in enumerate(paths):
        write_png(p, good, [k % 5])
    ingest, eng = pkg("ingest"), pkg("engine")
    staged = ingest.stage_png_batch([str(p) for p in paths], H, W, 2)
    assert (staged["status"] == 0).all()
    staged["channels"][1] = 0
    staged["channels"][2] = 9
    staged["channels"][3] = -3
    staged["length"][4] = -1000
    staged["length"][5] = 5
    # (a stream of more than twice the bytes it is to produce is declined as well: the stage arrays are given room for it)
    got = eng.png_inflate_batch(staged, H, W, gpu)
    status, gray = got["status"].cpu().numpy(), got["gray"].cpu().numpy()
    assert status[0] == 0 and status[6] == 0 and (status[1:6] != 0).all()
    np.testing.assert_array_equal(gray[0], good)
    np.testing.assert_array_equal(gray[6], good)


def test_many_files_in_one_call(gpu, tmp_path):
    """A few thousand views in one launch, against the host decoder (which equals PIL, tests/test_host.py)."""
    from PIL import Image
    rng = np.random.default_rng(11)
    H, W = 224, 224
    base = [view_like(rng, H, W) for _ in range(24)]
    paths = []
    for k in range(24):
        p = tmp_path / f"b{k}.png"
        Image.fromarray(base[k]).save(p)
        paths.append(p)
    many = [paths[int(i)] for i in rng.integers(0, 24, 3000)]
    staged, rows, gray, status = decode_on_device(many, H, W, gpu, workers=8)
    assert (status == 0).all()
    host = np.empty((len(many), H, W), dtype=np.uint8)
    st, _ = pkg("ingest").decode_png_batch([str(p) for p in many], H, W, host, 8)
    assert (st == 0).all()
    np.testing.assert_array_equal(gray, host)


def test_streams_of_a_few_bits_per_match(gpu, tmp_path):
    """Flat and periodic images of hundreds of kilobytes deflate to a few hundred bytes: thousands of maximal
    matches at two or three bits each, far more decoding rounds than stream bytes.  The device decodes them."""
    from PIL import Image
    H, W = 281, 711
    flat = np.full((H, W, 4), 200, dtype=np.uint8)
    period = np.resize(np.arange(7, dtype=np.uint8) * 30, (H, W, 3)).copy()
    paths = [tmp_path / "flat.png", tmp_path / "period.png", tmp_path / "flat9.png"]
    write_png(paths[0], flat, [0], level=9)
    write_png(paths[1], period, [0, 2], level=6)
    Image.fromarray(flat[:, :, 0]).save(paths[2], compress_level=9)
    assert all(p.stat().st_size < 4000 for p in paths)
    staged, rows, gray, status = decode_on_device(paths, H, W, gpu)
    assert (staged["status"] == 0).all() and (status == 0).all(), (staged["status"], status)
    for i, p in enumerate(paths):
        np.testing.assert_array_equal(gray[i], np.asarray(Image.open(p).convert("L")), err_msg=p.name)


def test_random_files_against_pil(gpu, tmp_path):
    """tools/pngfuzz.py's batches (random sizes, channels, filters, zlib levels, strategies and windows, IDAT splits,
    damaged streams), a fixed seed."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import pngfuzz
    rng = np.random.default_rng(20261112)
    for _ in range(20):
        pngfuzz.run_batch(rng, gpu, tmp_path)

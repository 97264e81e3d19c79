"""N3: on-disk formats around the path (finish.txt, source-view LUT, PNG quantisation, PSNR/SSIM, checkpoints)."""
import os

import numpy as np
import torch

import golden_util as gu
from pixel_nerf_multiscale_amd import evalio


def test_finish_log_resume_and_append(tmp_path):
    src = os.path.join(gu.GOLDEN_DIR, "finish_excerpt.txt")      # first 5 lines of the reference's eval/finish.txt
    p = tmp_path / "out" / "finish.txt"
    os.makedirs(p.parent)
    p.write_text(open(src).read() + "broken line\n")
    log = evalio.FinishLog(str(p))
    rows = [l.split() for l in open(src)]
    assert log.cnt == 5 and log.finished == {r[0] for r in rows}
    assert abs(log.mean()[0] - np.mean([float(r[1]) for r in rows])) < 1e-12
    log.append("newobj", 20.5, 0.8)
    log.close()
    again = evalio.FinishLog(str(p))
    assert again.cnt == 6 and "newobj" in again.finished
    assert open(p).read().splitlines()[-1] == "newobj 20.5 0.8 1"


def test_view_lists(tmp_path):
    f = tmp_path / "src.txt"
    f.write_text("03691459 101354f9d8dede686f7b08d9de913afe 20\n04256520 abc 1 5 9\n")
    lut = evalio.read_source_view_lut(str(f))
    assert lut["03691459/101354f9d8dede686f7b08d9de913afe"].tolist() == [20] and lut["04256520/abc"].tolist() == [1, 5, 9]
    g = tmp_path / "views.txt"
    g.write_text("0 3 7 11\nignored\n")
    assert evalio.read_eval_view_list(str(g)).tolist() == [0, 3, 7, 11]


def test_quantise_psnr_ssim():
    assert evalio.quantize_uint8(np.array([-0.1, 0.0, 0.5, 0.999, 1.0, 1.2])).tolist() == [0, 0, 127, 254, 255, 255]
    rng = np.random.default_rng(0)
    a = rng.random((32, 40, 3))
    assert evalio.ssim(a, a) == 1.0
    assert abs(evalio.psnr(a, a + 0.1) - 20.0) < 1e-9
    b = np.clip(a + rng.normal(0, 0.05, a.shape), 0, 1)
    s = evalio.ssim(a, b)
    assert 0.5 < s < 1.0 and abs(evalio.ssim(b, a) - s) < 1e-12


def test_checkpoint_schemas(tmp_path):
    from hip_util import model_conf
    from pixel_nerf_multiscale_amd import PixelNeRFNet
    spec = gu.CASES["tiny_ns1"]
    spec = dict(spec, lat=[(256, 4, 4)])          # default encoder width so the stock constructor's MLP shapes match
    src = PixelNeRFNet(model_conf(spec))
    with torch.no_grad():
        src.mlp_coarse.lin_out.bias.fill_(0.25)
    sd = src.state_dict()
    files = {"bare": sd, "trainer": {"epoch": 3, "net_state_dict": sd, "best_val_loss": 1.0},
             "rewrite": {"model_state_dict": sd}, "dp": {"model": {"module." + k: v for k, v in sd.items()}}}
    for name, obj in files.items():
        path = tmp_path / name
        torch.save(obj, path)
        dst = PixelNeRFNet(model_conf(spec))
        missing, unexpected = evalio.load_checkpoint(dst, str(path))
        assert not missing and not unexpected, name
        assert float(dst.mlp_coarse.lin_out.bias.detach()[0]) == 0.25, name


def test_write_png_round_trip(tmp_path):
    """evalio.write_png: a standard 8-bit RGB PNG (decoded here with zlib by hand: signature, IHDR, filter-0 scanlines)."""
    import struct
    import zlib
    import numpy as np
    from pixel_nerf_multiscale_amd import evalio
    rng = np.random.default_rng(0)
    img = rng.random((13, 7, 3)).astype(np.float32)
    u8 = evalio.quantize_uint8(img)
    p = tmp_path / "a.png"
    evalio.write_png(str(p), u8)
    raw = p.read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(raw):
        n, tag = struct.unpack(">I4s", raw[pos:pos + 8])
        data = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + data) & 0xFFFFFFFF
        chunks.append((tag, data))
        pos += 12 + n
    assert [c[0] for c in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    w, h, depth, ctype = struct.unpack(">IIBB", chunks[0][1][:10])
    assert (w, h, depth, ctype) == (7, 13, 8, 2)
    rows = zlib.decompress(chunks[1][1])
    back = np.frombuffer(rows, np.uint8).reshape(13, 1 + 7 * 3)
    assert (back[:, 0] == 0).all() and (back[:, 1:].reshape(13, 7, 3) == u8).all()

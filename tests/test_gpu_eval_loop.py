"""N3: the per-object evaluation loop (reference eval/eval.py:186-362) on a synthetic two-object dataset.  Ground-truth
images are this package's own fp32-path renders of the same scene (no dataset exists here), so the loop's PSNR must come
out at the low-precision kernel's level; what is really under test is the driver logic: source / target view selection,
encode -> render -> clamp -> quantise -> metrics, finish.txt lines and resume."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _Objects(list):
    z_near, z_far, lindisp = 1.25, 2.75, False


def _make_dataset(net32, rend, n_obj, NV, W, H, focal, seed=777):
    """Ground truth = this package's fp32-path render of every target view with the jitter evaluate(seed=seed) will draw for
    that (object, view): evalio keys the per-view seed by frame_seed(frame_seed(seed, object), view)."""
    import golden_util as gu
    from pixel_nerf_multiscale_amd.parallel import frame_seed
    data = _Objects()
    for o in range(n_obj):
        poses = torch.from_numpy(np.stack([gu.pose_spherical(40.0 * v + 13.0 * o, -20.0 - 3.0 * o, 2.0) for v in range(NV)]))
        g = torch.Generator().manual_seed(100 + o)
        src_img = torch.rand(1, 3, H, W, generator=g) * 2 - 1                 # the source view is a random image: only the
        images = torch.zeros(NV, 3, H, W)                                       # trunk sees it
        images[0] = src_img[0]
        net32.encode(src_img.cuda()[None], poses[:1].cuda()[None], torch.tensor(focal)[None].cuda())
        for v in range(1, NV):
            rend.forced_seed = frame_seed(frame_seed(seed, o), v)               # the same jitter in ground truth and evaluation
            rgb, _ = rend.render_image(net32, poses[v], W, H, focal, data.z_near, data.z_far)
            images[v] = (rgb.clamp(0, 1).permute(2, 0, 1) * 2 - 1).cpu()
        data.append(dict(path=f"/data/cat{o % 2}/obj{o:03d}", images=images, poses=poses, focal=focal))
    rend.forced_seed = None
    return data


def test_evaluate_two_objects_resume_and_lut(tmp_path):
    import golden_util as gu
    from hip_util import model_conf
    from pixel_nerf_multiscale_amd import NeRFRenderer, PixelNeRFNet, evalio
    spec = dict(gu.CASES["full_ns1"])
    W = H = 32
    focal, NV = 33.0, 4
    torch.manual_seed(0)
    net = PixelNeRFNet(model_conf(spec, "fp32")).cuda().eval()
    for which, mlp in (("coarse", net.mlp_coarse), ("fine", net.mlp_fine)):
        mlp.load_state_dict({k: torch.from_numpy(v) for k, v in gu.make_mlp_state(spec, which).items()})
    rend = NeRFRenderer(n_coarse=32, n_fine=16, n_fine_depth=8, white_bkgd=True).cuda().eval()
    data = _make_dataset(net, rend, 2, NV, W, H, focal)
    out = str(tmp_path / "eval_out")

    # pass 1: only the first object (max_objects = 1), fixed source view "0", fp16 kernel
    net.precision = "fp16"
    m1 = evalio.evaluate(net, rend, data, out, source="0", max_objects=1, verbose=False, seed=777)
    lines = open(os.path.join(out, "finish.txt")).read().split("\n")
    assert len([x for x in lines if x]) == 1 and lines[0].split()[0] == "obj000" and lines[0].split()[3] == "1"
    assert m1[2] == 1 and 45.0 <= m1[0] < 99.0 and 0.99 <= m1[1] <= 1.0            # fp16 render vs fp32 ground truth
    pngs = sorted(os.listdir(os.path.join(out, "obj000")))
    assert pngs == ["000001.png", "000002.png", "000003.png"]                      # the source view 0 is not a target

    # pass 2: resume — object 0 is skipped (no new line for it), object 1 is appended; source views from a look-up table
    lut = tmp_path / "src.txt"
    lut.write_text("cat0 obj000 0\ncat1 obj001 0\n")
    calls = []
    orig = rend.render_image
    rendered = []

    def recording(*a, **k):
        calls.append(1)
        res = orig(*a, **k)
        rendered.append(res[0].detach().clone())
        return res
    rend.render_image = recording
    m2 = evalio.evaluate(net, rend, data, out, viewlist=str(lut), verbose=False, seed=777)
    rend.render_image = orig
    rows = [x.split() for x in open(os.path.join(out, "finish.txt")).read().split("\n") if x]
    assert [r[0] for r in rows] == ["obj000", "obj001"] and len(calls) == NV - 1   # only object 1 was rendered
    assert m2[2] == 2
    want = (float(rows[0][1]) + float(rows[1][1])) / 2, (float(rows[0][2]) + float(rows[1][2])) / 2
    assert abs(m2[0] - want[0]) < 1e-9 and abs(m2[1] - want[1]) < 1e-9              # the written lines are the running means' terms
    assert abs(float(rows[0][1]) - m1[0]) < 1e-9

    # pass 3: everything finished -> nothing rendered, same means; eval view list + include_src + multicat names
    calls.clear()
    rend.render_image = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    m3 = evalio.evaluate(net, rend, data, out, source="0", verbose=False, seed=777)
    assert not calls and m3 == m2
    out2 = str(tmp_path / "eval_out2")
    m4 = evalio.evaluate(net, rend, data, out2, source="0", eval_view_list=[0, 2], include_src=True, multicat=True,
                         no_compare_gt=True, verbose=False)
    rend.render_image = orig
    assert sorted(os.listdir(out2)) == ["cat0_obj000", "cat1_obj001", "finish.txt"]
    assert sorted(os.listdir(os.path.join(out2, "cat1_obj001"))) == ["000000.png", "000002.png"]
    assert m4 == (0.0, 0.0, 2)

    # the PNGs hold the truncating quantisation of the clamped render (eval.py:301)
    import struct, zlib
    raw = open(os.path.join(out, "obj001", "000002.png"), "rb").read()
    idat = raw[raw.index(b"IDAT") + 4:]
    n = struct.unpack(">I", raw[raw.index(b"IDAT") - 4:raw.index(b"IDAT")])[0]
    px = np.frombuffer(zlib.decompress(idat[:n]), np.uint8).reshape(H, 1 + 3 * W)[:, 1:].reshape(H, W, 3)
    # the frame evaluate() rendered for that file (object 1's targets are views 1, 2, 3: the second call of pass 2) — not a
    # second encode + render: the trunk goes through MIOpen, whose algorithm choice (and with it the latent's last bits) may
    # differ from call to call
    want = evalio.quantize_uint8(rendered[1].clamp(0, 1).reshape(H, W, 3).cpu().numpy())
    assert np.array_equal(px, want)

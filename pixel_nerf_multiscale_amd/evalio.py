"""
On-disk / wire formats around the render path, so an evaluation driver built on this package reads
and writes what the reference's eval scripts do.  Pure host-side parsing and arithmetic.

* finish.txt resume log      reference eval/eval.py:113-133,360-362   one line per object: "<name> <psnr> <ssim> <cnt>"
* source-view look-up table   reference eval/eval.py:156-165           viewlist/src_*.txt: "<cat> <obj> <view> [<view> ...]"
* eval view list              reference eval/eval.py:170-176           first line: target view indices
* PNG quantisation            reference eval/eval.py:301               (rgb * 255).astype(uint8) after clamp to [0,1]
* PSNR / SSIM                 reference eval/eval.py:324-332           skimage.measure.compare_psnr / compare_ssim
                              (data_range=1, multichannel): restated from the published definitions — skimage is not
                              installed here and the reference's own numbers need its dataset, so SSIM is "parity unpinned".
* checkpoints                 three mutually incompatible schemas in the reference tree
* evaluate()                  reference eval/eval.py:186-362           the per-object loop that composes the above with
                              encode -> render -> clamp -> metrics -> finish.txt (resume), on this package's renderer
"""
import os

import numpy as np
import torch


class FinishLog:
    """finish.txt: append-only per-object results with resume (objects already listed are skipped by the driver)."""

    def __init__(self, path, state=None):
        """state = (finished, total_psnr, total_ssim, cnt): a follower of a multi-rank evaluation — it starts from the
        owner's view of the file (broadcast once) and never opens it: the file has ONE writer."""
        self.path = path
        self.finished, self.total_psnr, self.total_ssim, self.cnt = set(), 0.0, 0.0, 0
        self._f = None
        if state is not None:
            self.finished, self.total_psnr, self.total_ssim, self.cnt = set(state[0]), float(state[1]), float(state[2]), int(state[3])
            return
        if os.path.exists(path):
            with open(path) as f:
                rows = [x.strip().split() for x in f.readlines()]
            rows = [x for x in rows if len(x) == 4]
            self.finished = {x[0] for x in rows}
            self.total_psnr = sum(float(x[1]) for x in rows)
            self.total_ssim = sum(float(x[2]) for x in rows)
            self.cnt = sum(int(x[3]) for x in rows)
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        self._f = open(path, "a", buffering=1)

    def state(self):
        return (sorted(self.finished), self.total_psnr, self.total_ssim, self.cnt)

    def append(self, obj_name, psnr, ssim, cnt=1):
        if self._f is not None:
            self._f.write("{} {} {} {}\n".format(obj_name, psnr, ssim, cnt))
        self.finished.add(obj_name)
        self.total_psnr += psnr
        self.total_ssim += ssim
        self.cnt += cnt

    def mean(self):
        return (self.total_psnr / self.cnt, self.total_ssim / self.cnt) if self.cnt else (0.0, 0.0)

    def close(self):
        if self._f is not None:
            self._f.close()


def read_source_view_lut(path):
    """{"<cat>/<obj>": LongTensor(source view indices)}"""
    lut = {}
    with open(path) as f:
        for line in f:
            x = line.strip().split()
            if len(x) >= 3:
                lut[x[0] + "/" + x[1]] = torch.tensor(list(map(int, x[2:])), dtype=torch.long)
    return lut


def read_eval_view_list(path):
    with open(path) as f:
        return torch.tensor(list(map(int, f.readline().split())), dtype=torch.long)


def quantize_uint8(rgb):
    """float image in [0,1] (clamped) -> uint8 the way the reference writes PNGs (truncation, not rounding)."""
    a = np.clip(np.asarray(rgb, dtype=np.float32), 0.0, 1.0)
    return (a * 255).astype(np.uint8)


def psnr(img, gt, data_range=1.0):
    err = np.mean((np.asarray(img, np.float64) - np.asarray(gt, np.float64)) ** 2)
    return 10.0 * np.log10((data_range ** 2) / err)


def ssim(img, gt, data_range=1.0, win_size=7, K1=0.01, K2=0.03):
    """Mean structural similarity, uniform 7x7 window, sample covariance, per channel then averaged, borders cropped
    (Wang et al. 2004 as implemented by skimage.measure.compare_ssim with multichannel=True)."""
    from scipy.ndimage import uniform_filter
    x, y = np.asarray(img, np.float64), np.asarray(gt, np.float64)
    if x.ndim == 3:
        return float(np.mean([ssim(x[..., c], y[..., c], data_range, win_size, K1, K2) for c in range(x.shape[-1])]))
    NP = win_size ** 2
    cov_norm = NP / (NP - 1.0)
    ux, uy = uniform_filter(x, win_size), uniform_filter(y, win_size)
    uxx, uyy, uxy = uniform_filter(x * x, win_size), uniform_filter(y * y, win_size), uniform_filter(x * y, win_size)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    pad = (win_size - 1) // 2
    return float(S[pad:-pad, pad:-pad].mean())


def load_checkpoint(net, path, device=None, strict=False):
    """Load any of the reference tree's three checkpoint schemas into a PixelNeRFNet, executing nothing
    from the file (weights_only):  a bare state-dict (models.py.backup2:293-305, upstream pixelNeRF);
    {"net_state_dict": ...} (train/trainlib/trainer.py:593-600);  {"model_state_dict"| "model": ...} (models.py:331-336).
    Returns the (missing, unexpected) key lists of load_state_dict."""
    ck = torch.load(path, map_location=device or "cpu", weights_only=True)
    for key in ("net_state_dict", "model_state_dict", "model", "state_dict"):
        if isinstance(ck, dict) and key in ck and isinstance(ck[key], dict):
            ck = ck[key]
            break
    ck = {k[len("module."):] if k.startswith("module.") else k: v for k, v in ck.items()}   # DataParallel prefix
    res = net.load_state_dict(ck, strict=strict)
    return list(res.missing_keys), list(res.unexpected_keys)


def write_png(path, rgb_u8):
    """8-bit RGB PNG from an (H, W, 3) uint8 array with the standard library only (the reference writes through imageio,
    eval/eval.py:301; the pixel values are what matters for calc_metrics.py, not the encoder)."""
    import struct
    import zlib
    a = np.ascontiguousarray(rgb_u8, dtype=np.uint8)
    h, w, c = a.shape
    assert c == 3
    raw = b"".join(b"\x00" + a[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def evaluate(net, renderer, dataset, output_dir="", *, source="", viewlist=None, eval_view_list=None,
             include_src=False, scale=1.0, multicat=False, gpu_id=None, ray_batch_size=50000, no_compare_gt=False,
             write_compare=False, write_images=True, max_objects=50, z_near=None, z_far=None,
             verbose=True, seed=None):
    """The per-object evaluation loop of the reference (eval/eval.py:186-362) on this package's renderer.

    dataset: a sequence of per-object dicts as the reference's datasets yield them (unbatched): "path", "images"
    (NV, 3, H, W) in [-1, 1], "poses" (NV, 4, 4) camera-to-world, "focal" (float | tensor), optional "c"; z_near / z_far
    from the arguments or the dataset's attributes (eval.py:153-154).  net / renderer as the driver sets them up
    (eval.py:136-151: caller-side overrides of n_coarse / n_fine / mlp_fine stay with the caller).

    Per object (eval.py:186-362): resume — objects already in <output_dir>/finish.txt are skipped (:113-133, :203-205);
    source views from `source` ("0 1 2") or the look-up table file / dict `viewlist` keyed "<cat>/<obj>" (:156-165,
    :224-231); target views = eval_view_list (file / indices) minus the source views unless include_src (:170-178,
    :246-248); net.encode on the source images (:271-276); every target view rendered — as ONE call per view with the rays
    generated inside the render launch (NeRFRenderer.render_image) on one device, or, when a process group is up, through the
    sharded form of renderer.bind_parallel(net, gpu_id, simple_output=True) in chunks of ray_batch_size (:151, :267,
    :279-284: every chunk cut over the ranks, one all_gather each; rank 0 alone writes finish.txt and the PNGs, the other
    ranks take the resume state from it by one broadcast) — with the
    frames brought to the host asynchronously (frame_to_host_async) while the next view renders; clamp to [0, 1] and
    reshape (:286-293); PNGs "<obj>/<view:06>.png" quantised by truncation (:294-301); PSNR / SSIM per view against
    images * 0.5 + 0.5, averaged per object (:318-347); running means and a "<obj> <psnr> <ssim> 1" line appended to
    finish.txt (:348-362).  Returns (mean_psnr, mean_ssim, n_objects_counted) over everything in finish.txt.

    Deliberate differences: the rays of all target views are not concatenated and re-split (:250-267) — a view is the unit;
    the random jitter is keyed by (seed, ray): one base seed per call (`seed`, else drawn from torch's generator on rank 0
    and broadcast) and a seed derived per (object, view), so neither the chunk size, nor the number of ranks, nor a resume changes a pixel; SSIM is this module's restatement
    (skimage is not importable here: parity unpinned); depth EXR / colour-mapped depth outputs (:303-316) are not written."""
    import torch.distributed as dist
    from . import util
    dev = net.poses.device
    z_near = float(getattr(dataset, "z_near", None) if z_near is None else z_near)
    z_far = float(getattr(dataset, "z_far", None) if z_far is None else z_far)
    has_output = bool(output_dir and str(output_dir).strip())
    sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank = dist.get_rank() if sharded else 0
    # Under a process group every rank runs this loop (each call's rays are cut over the ranks and every rank gets the whole
    # frame back), but the output directory has ONE writer: rank 0 owns finish.txt and the PNGs; the other ranks start from
    # rank 0's view of the file — broadcast once, so all ranks skip the same objects whatever the file system shows them —
    # and keep their running means in memory.  The return value is the same on every rank.
    log = None
    if has_output:
        if rank == 0:
            log = FinishLog(os.path.join(output_dir, "finish.txt"))
        if sharded:
            box = [log.state() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            if rank != 0:
                log = FinishLog(os.path.join(output_dir, "finish.txt"), state=box[0])
    writes_files = has_output and rank == 0
    from .parallel import frame_seed
    base_seed = int(seed) if seed is not None else util.seed_from_torch()
    if sharded and seed is None:
        box = [base_seed]
        dist.broadcast_object_list(box, src=0)
        base_seed = int(box[0])
    total_psnr, total_ssim, cnt = (log.total_psnr, log.total_ssim, log.cnt) if log else (0.0, 0.0, 0)
    if log and log.cnt > 0 and verbose:
        print("resume psnr", log.total_psnr / log.cnt, "ssim", log.total_ssim / log.cnt)

    if isinstance(viewlist, str) and viewlist:
        viewlist = read_source_view_lut(viewlist)
    use_lut = bool(viewlist)
    fixed_source = None if use_lut else torch.tensor(sorted(int(x) for x in str(source).split()), dtype=torch.long)
    if isinstance(eval_view_list, str):
        eval_view_list = read_eval_view_list(eval_view_list)
    render_par = None
    if sharded:
        # built directly: bind_parallel() only shards when the caller lists several gpu ids (eval.py:151 passes args.gpu_id),
        # and a rank of a one-process-per-GPU job has exactly one
        from .render.nerf import _ShardedRenderWrapper
        render_par = _ShardedRenderWrapper(net, renderer, simple_output=True).eval()
    was_training = net.training
    net.eval()
    try:
        with torch.no_grad():
            for obj_idx in range(len(dataset)):
                if max_objects is not None and obj_idx >= max_objects:      # eval.py:187 stops after 50 objects
                    break
                data = dataset[obj_idx]
                dpath = data["path"]
                obj_base, cat_name = os.path.basename(dpath), os.path.basename(os.path.dirname(dpath))
                obj_name = cat_name + "_" + obj_base if multicat else obj_base
                if verbose:
                    print("OBJECT", obj_idx, "OF", len(dataset), dpath)
                if log and obj_name in log.finished:
                    if verbose:
                        print("(skip)")
                    continue
                images = data["images"]                                       # (NV, 3, H, W)
                NV, _, H, W = images.shape
                if scale != 1.0:
                    Ht, Wt = int(H * scale), int(W * scale)
                    if abs(Ht / scale - H) > 1e-10 or abs(Wt / scale - W) > 1e-10:
                        import warnings
                        warnings.warn(f"Inexact scaling, please check {scale} times ({H}, {W}) is integral")
                    H, W = Ht, Wt
                src = viewlist[cat_name + "/" + obj_base] if use_lut else fixed_source
                src_mask = torch.zeros(NV, dtype=torch.bool)
                src_mask[src] = True
                tgt_mask = torch.ones(NV, dtype=torch.bool)
                if eval_view_list is not None:
                    tgt_mask = torch.zeros(NV, dtype=torch.bool)
                    tgt_mask[torch.as_tensor(eval_view_list, dtype=torch.long)] = True
                if not include_src:
                    tgt_mask = tgt_mask & ~src_mask
                novel = tgt_mask.nonzero(as_tuple=False).reshape(-1)
                focal = data["focal"]
                focal = torch.tensor(focal, dtype=torch.float32) if isinstance(focal, float) else torch.as_tensor(focal).float()
                c = data.get("c")
                c = None if c is None else torch.as_tensor(c).float()
                poses = torch.as_tensor(data["poses"]).float()
                net.encode(images[src_mask].to(dev).unsqueeze(0), poses[src_mask].to(dev).unsqueeze(0), focal[None].to(dev),
                           c=None if c is None else c.to(dev).unsqueeze(0))
                if sharded:
                    # every rank ran the trunk on the same images, but a convolution library may pick different algorithms in
                    # different processes (last-bit differences): rank 0's maps are THE maps, so that the gathered frame is one
                    # consistent render (64 KB ... 6 MB per object, once — SURVEY 8e)
                    maps = [m.detach().clone() for m in net.encoder.level_maps()]
                    for m in maps:
                        if dist.get_backend() == "nccl":
                            dist.broadcast(m, src=0)
                        else:
                            h = m.cpu()
                            dist.broadcast(h, src=0)
                            m.copy_(h)
                    net.encoder.set_latents(maps)
                frames, pending = [], None
                for vi in novel.tolist():
                    # keyed by (object, view), not by a running count: a resumed run draws what the uninterrupted run drew
                    view_seed = frame_seed(frame_seed(base_seed, obj_idx), vi)
                    if render_par is None:
                        keep_seed, renderer.forced_seed = renderer.forced_seed, view_seed
                        try:
                            rgb, depth = renderer.render_image(net, poses[vi], W, H, focal * scale, z_near, z_far,
                                                               c=None if c is None else c * scale)
                        finally:
                            renderer.forced_seed = keep_seed
                    else:
                        rays = util.gen_rays_device(poses[vi], W, H, focal * scale, z_near, z_far,
                                                    c=None if c is None else c * scale, device=dev)
                        parts, at = [], 0
                        for r in torch.split(rays, ray_batch_size, dim=0):
                            parts.append(render_par(r[None], ray_index_base=at, seed=view_seed))
                            at += r.shape[0]
                        rgb = torch.cat([p[0][0] for p in parts], 0).reshape(H, W, 3)
                        depth = torch.cat([p[1][0] for p in parts], 0).reshape(H, W)
                    nxt = renderer.frame_to_host_async(rgb, depth)          # D2H overlaps the next view's render
                    if pending is not None:
                        pending[2].synchronize()
                        frames.append(pending[0])
                    pending = nxt
                if pending is not None:
                    pending[2].synchronize()
                    frames.append(pending[0])
                all_rgb = torch.clamp(torch.stack(frames), 0.0, 1.0).numpy() if frames else np.zeros((0, H, W, 3), np.float32)
                n_gen = len(frames)
                if writes_files and write_images:
                    obj_out = os.path.join(output_dir, obj_name)
                    os.makedirs(obj_out, exist_ok=True)
                    for i in range(n_gen):
                        write_png(os.path.join(obj_out, "{:06}.png".format(int(novel[i]))), quantize_uint8(all_rgb[i]))
                curr_psnr = curr_ssim = 0.0
                if not no_compare_gt and n_gen:
                    gt = (images * 0.5 + 0.5)[tgt_mask].permute(0, 2, 3, 1).contiguous().numpy()
                    for i in range(n_gen):
                        curr_ssim += ssim(all_rgb[i], gt[i], data_range=1)
                        curr_psnr += psnr(all_rgb[i], gt[i], data_range=1)
                        if writes_files and write_compare:
                            write_png(os.path.join(output_dir, obj_name, "{:06}_compare.png".format(int(novel[i]))),
                                      quantize_uint8(np.hstack((all_rgb[i], gt[i]))))
                    curr_psnr /= n_gen
                    curr_ssim /= n_gen
                total_psnr += curr_psnr
                total_ssim += curr_ssim
                cnt += 1
                if verbose and not no_compare_gt:
                    print("curr psnr", curr_psnr, "ssim", curr_ssim, "running psnr", total_psnr / cnt, "running ssim", total_ssim / cnt)
                if log:
                    log.append(obj_name, curr_psnr, curr_ssim, 1)
    finally:
        net.train(was_training)
        if log:
            log.close()
    if sharded:
        dist.barrier()              # rank 0's files are complete when any rank returns
    if verbose and cnt:
        print("final psnr", total_psnr / cnt, "ssim", total_ssim / cnt)
    return (total_psnr / cnt, total_ssim / cnt, cnt) if cnt else (0.0, 0.0, 0)

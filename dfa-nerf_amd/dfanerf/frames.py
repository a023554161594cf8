"""Training input stage on the device.

Upstream decodes two JPEGs from disk every optimisation step (run_nerf_com_trainExpLater.py:771-774: imageio.imread of
the head and the composite ground truth), uploads both 450x450 frames and picks 2048 pixels of each (:791-800) after
drawing them with NumPy on the host (:786-820).  At a ~2 ms GPU step that host work is the step.  Here:

  * DeviceFrameCache - the ground-truth frames live on the device as uint8 [H*W,3] (607 KB per 450x450 frame; a
    7000-frame sequence x 2 image sets = 8.5 GB of the 288 GB): decoded once (thread pool) or, when the sequence does
    not fit the budget, kept in an LRU of decoded frames.  After the first visit of a frame a step reads no file and
    copies nothing from the host.
  * PixelSampler - the pixel draw of MAIN:786-820 on the device: N_rand DISTINCT pixels, uniform over the subsets
    (np.random.choice(replace=False) upstream), with the face-rect / lower-half split when sample_rate > 0; one random
    key per pixel and a top-k, no host round trip.
  * the targets are never materialised: dfn_mse_loss_u8 (training.MseLossFn) gathers them from the uint8 frames inside
    the loss kernel."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch


def _imread_u8(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert('RGB'), dtype=np.uint8)


class DeviceFrameCache:
    """uint8 ground-truth frames on the device: get(i) -> (head [H*W,3], com [H*W,3]) uint8 device tensors.

    budget_bytes: device memory the cache may use.  If every frame of `ids` fits, preload() decodes them all up front;
    otherwise frames are decoded on first use and the least recently used ones are dropped."""

    def __init__(self, paths_head, paths_com, H, W, device, budget_bytes=32 << 30, reader=_imread_u8):
        self.paths = (list(paths_head), list(paths_com))
        self.H, self.W, self.device, self.reader = int(H), int(W), device, reader
        self.frame_bytes = 2 * self.H * self.W * 3
        self.capacity = max(1, int(budget_bytes // self.frame_bytes))
        self.slots = {}                 # frame id -> (head, com)
        self.order = []                 # LRU order (oldest first); only maintained once the cache is full
        self.host_reads = 0             # files decoded so far (tools/profile_train_host.py reports it per step)
        self._pin = None

    def _decode(self, i):
        return self.reader(self.paths[0][i]), self.reader(self.paths[1][i])

    def _upload(self, i, head, com):
        n = self.H * self.W * 3
        for a in (head, com):
            if a.shape != (self.H, self.W, 3):
                raise ValueError(f"frame {i}: image shape {a.shape}, expected {(self.H, self.W, 3)}")
        if self._pin is None:
            self._pin = [torch.empty(2 * n, dtype=torch.uint8, pin_memory=torch.cuda.is_available()) for _ in range(2)]
            self._pin_ev = [None, None]
            self._pin_k = 0
        k = self._pin_k
        self._pin_k ^= 1
        if self._pin_ev[k] is not None:
            self._pin_ev[k].synchronize()
        buf = self._pin[k]
        buf[:n].numpy()[:] = head.reshape(-1)
        buf[n:].numpy()[:] = com.reshape(-1)
        dev = buf.to(self.device, non_blocking=True)
        if dev.data_ptr() == buf.data_ptr():       # the cache lives on the staging buffer's own device (CPU unit tests)
            dev = buf.clone()
        if torch.cuda.is_available():
            self._pin_ev[k] = torch.cuda.Event()
            self._pin_ev[k].record()
        self.host_reads += 2
        pair = (dev[:n].view(-1, 3), dev[n:].view(-1, 3))
        if len(self.slots) >= self.capacity:
            old = self.order.pop(0)
            del self.slots[old]
        self.slots[i] = pair
        self.order.append(i)
        return pair

    def preload(self, ids, workers=8, log=None):
        """Decode and upload `ids` (as many as fit the budget) with a pool of decoder threads."""
        ids = [int(i) for i in ids if int(i) not in self.slots][:self.capacity - len(self.slots)]
        with ThreadPoolExecutor(max_workers=workers) as pool:
            for k, (i, (h, c)) in enumerate(zip(ids, pool.map(self._decode, ids))):
                self._upload(i, h, c)
                if log and (k + 1) % 500 == 0:
                    log(f"[dfanerf] ground-truth frames on the device: {k + 1}/{len(ids)}")
        return len(ids)

    def get(self, i):
        i = int(i)
        hit = self.slots.get(i)
        if hit is None:
            return self._upload(i, *self._decode(i))
        if len(self.slots) >= self.capacity and self.order and self.order[-1] != i:      # LRU bookkeeping only when full
            self.order.remove(i)
            self.order.append(i)
        return hit


class PixelSampler:
    """MAIN:786-820 on the device.  draw(...) -> int32 [N_rand] pixel ids y*W+x, all distinct:
      sample_rate == 0: a uniformly random N_rand-subset of the H*W pixels in random order;
      sample_rate > 0:  int(N_rand * sample_rate) of them from (face rect | lower half of the image), the rest from the
                        complement.  The face rectangle [y0, x0, h, w] comes from `rects` (host array [frames, 4], uploaded
                        once) by frame index, or is passed to draw() directly.
    On the GPU one launch of dfn_sample_pixels (rejection sampling over 8192 counter-based candidates, an LDS hash table
    for the duplicates, a block scan for the order); when a class is too small for that (tiny images, almost every pixel
    requested) and on the CPU: one uniform key per pixel + top-k in torch.  Both are exactly uniform over the subsets up
    to the generator, like np.random.choice(replace=False)."""
    CANDIDATES = 8192            # SAMPLE_PIXELS_CANDIDATES of the kernel

    def __init__(self, H, W, n_rand, sample_rate, device, seed=0, rects=None, pipeline=False, stream=None):
        # pipeline: draw on a side stream into a ring of three buffers - the draw of step n + 1 then runs while the main
        # stream still works on step n instead of in front of its forward (20 us + a launch gap of a 1.9-ms step).  A
        # returned tensor is overwritten by the third draw after it: for training loops that consume a draw within its
        # own step (run_nerf.train, bench.py), not for callers that collect draws.
        # stream: the side stream to use (default: a new one).  The device runs four hardware queues: a fifth stream shares
        # one of them with another stream and serialises with it - the training loop passes one of the conditioning
        # networks' streams (training.SignalTrainer.pose_stream()).
        self.pipeline, self._ring, self._done, self._k, self._side = bool(pipeline), None, [None] * 3, 0, stream
        self.H, self.W, self.n, self.rate, self.device = int(H), int(W), int(n_rand), float(sample_rate), device
        if self.n > self.H * self.W:
            raise ValueError("PixelSampler: more rays than pixels")
        self.seed, self.counter = int(seed) & ((1 << 63) - 1), 0
        self.rect_num = int(self.n * self.rate) if self.rate > 0 else 0
        self.rects_host = None if rects is None else np.asarray(rects, dtype=np.int64).reshape(-1, 4)
        self.rects_dev = None if rects is None else torch.as_tensor(self.rects_host, dtype=torch.int32, device=device)
        self.on_gpu = torch.device(device).type == "cuda"
        self._torch = None
        self.out = torch.empty(self.n, dtype=torch.int32, device=device) if self.on_gpu else None

    # ---- which path -----------------------------------------------------------------------------------------------
    def _inside_area(self, rect):
        """pixels of (rect | lower half): rows y >= H / 2 entirely, plus the part of the rectangle above them"""
        H, W = self.H, self.W
        half = -(-H // 2)                                        # first row with y >= H / 2
        y0, x0, h, w = [int(v) for v in rect]
        ya, yb = max(y0, 0), min(y0 + h, half - 1, H - 1)
        xa, xb = max(x0, 0), min(x0 + w, W - 1)
        above = max(0, yb - ya + 1) * max(0, xb - xa + 1)
        return (H - half) * W + above

    def _kernel_ok(self, rect_host):
        HW = self.H * self.W
        if not self.on_gpu or HW > 0x7fffffff or self.n > self.CANDIDATES // 2:
            return False
        frac = 1.0 - np.exp(-self.CANDIDATES / HW)               # expected share of a class's pixels among the candidates
        # the distinct candidates that land in a class are (nearly) Poisson with mean m = area * frac: ask for six standard
        # deviations of head-room, so that a shortfall (the kernel would leave output slots unwritten) is a < 1e-9 event
        enough = lambda m, need: m - 6.0 * np.sqrt(max(m, 0.0)) >= need
        if self.rect_num == 0:
            return enough(HW * frac, self.n)
        if rect_host is None:
            return False
        a_in = self._inside_area(rect_host)
        return enough(a_in * frac, self.rect_num) and enough((HW - a_in) * frac, self.n - self.rect_num)

    # ---- the torch path (CPU tests, tiny images) ----------------------------------------------------------------------
    def _draw_torch(self, rect):
        if self._torch is None:
            gen = torch.Generator(device=self.device)
            gen.manual_seed(self.seed)
            p = torch.arange(self.H * self.W, device=self.device)
            self._torch = (gen, torch.empty(self.H * self.W, dtype=torch.float32, device=self.device),
                           (p // self.W).to(torch.int32), (p % self.W).to(torch.int32))
        gen, keys, y, x = self._torch
        k = keys.uniform_(0.0, 1.0, generator=gen)
        if self.rect_num == 0:
            return torch.topk(k, self.n, sorted=True).indices.to(torch.int32)
        r = rect if isinstance(rect, torch.Tensor) else torch.as_tensor(np.asarray(rect), device=self.device)
        r = r.to(device=self.device, dtype=torch.int32)
        inside = ((y >= r[0]) & (y <= r[0] + r[2]) & (x >= r[1]) & (x <= r[1] + r[3])) | (y.float() >= self.H / 2)
        # keys of the other class drop below every real key: top-k never crosses the class boundary
        a = torch.topk(torch.where(inside, k, k - 2.0), self.rect_num, sorted=True).indices
        b = torch.topk(torch.where(inside, k - 2.0, k), self.n - self.rect_num, sorted=True).indices
        return torch.cat((a, b)).to(torch.int32)

    def draw(self, rect=None, frame=None):
        """rect: [y0, x0, h, w] (host sequence or tensor) or frame: index into `rects`; neither when sample_rate == 0."""
        rect_host = rect_dev = None
        if self.rect_num > 0:
            if frame is not None:
                rect_host, rect_dev = self.rects_host[int(frame)], self.rects_dev[int(frame)]
            elif isinstance(rect, torch.Tensor):
                rect_dev = rect.to(device=self.device, dtype=torch.int32)
                rect_host = None if rect.is_cuda else rect.cpu().numpy()
            else:
                rect_host = np.asarray(rect)
        if not self._kernel_ok(rect_host):
            return self._draw_torch(rect_dev if rect_dev is not None else rect_host)
        import ctypes as C
        from ._lib import check, lib
        if rect_dev is None and self.rect_num > 0:
            rect_dev = torch.as_tensor(np.asarray(rect_host, dtype=np.int32), device=self.device)
        self.counter += 1
        rect_ptr = None if rect_dev is None else C.c_void_p(rect_dev.contiguous().data_ptr())
        main = torch.cuda.current_stream()
        if self.pipeline and (self.rect_num == 0 or frame is not None):
            if self._ring is None:
                self._ring = [torch.zeros_like(self.out) for _ in range(3)]      # (a valid pixel id even if a slot were ever left unwritten)
                if self._side is None:
                    self._side = torch.cuda.Stream(device=self.device)
                self._side.wait_stream(main)
            k = self._k = (self._k + 1) % 3
            # everything enqueued on the main stream so far = the whole step that consumed the previous draw: once that is
            # done, slot k - 1 may be overwritten (by the draw after next); slot k waits for the step that used IT
            ev = torch.cuda.Event()
            ev.record(main)
            self._done[(k - 1) % 3] = ev
            if self._done[k] is not None:
                self._side.wait_event(self._done[k])
            out, stream = self._ring[k], self._side
        else:
            out, stream = torch.zeros_like(self.out), main          # a fresh tensor per draw: the previous one may still be in use
        check(lib.dfn_sample_pixels(self.H, self.W, self.n, self.rect_num, rect_ptr, C.c_uint64(self.seed),
                                    C.c_uint64(self.counter), C.c_void_p(out.data_ptr()), None,
                                    C.c_void_p(stream.cuda_stream)), "dfn_sample_pixels")
        if stream is not main:
            main.wait_stream(stream)
        return out

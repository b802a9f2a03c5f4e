"""Host side of the detect post-processing: device tensors in, device tensors / BoundBox out.

Mirrors the tail of FaceDetector.detect (reference face_detection.py:900-949) and the BoundBox
record (reference yolov3_detect.py:126-163)."""
import numpy as np

from ._lib import lib, ptr


class BoundBox(object):
    """Mutable detection record; same attribute surface as the reference's BoundBox
    (yolov3_detect.py:126-163): callers mutate xmin..ymax in place (face_detection.py:700-710)."""

    def __init__(self, xmin, ymin, xmax, ymax, objness=None, classes=None, anchor=None, subject_id=-1):
        self.xmin = xmin
        self.ymin = ymin
        self.xmax = xmax
        self.ymax = ymax
        self.objness = objness
        self.classes = classes
        self.anchor = anchor
        self.subject_id = subject_id
        self.label = -1
        self.score = -1

    def get_label(self):
        if self.label == -1:
            self.label = int(np.argmax(self.classes))
        return self.label

    def get_score(self):
        if self.score == -1:
            self.score = self.classes[self.get_label()]
        return np.min([self.score, 1.0])

    def get_relative_bb(self, width, height):
        return (int(self.xmin / width * 100.), int(self.ymin / height * 100.),
                int((self.xmax - self.xmin) / width * 100.), int((self.ymax - self.ymin) / height * 100.))


def decode_nms(ctx, head, image_size, conf_th, iou_th, num_cands):
    """head: (n, G, G, 6) float32 CUDA tensor -> dict of CUDA tensors
    boxes (n,K,4) i32, cell (n,K) i32, obj (n,K) f32, score (n,K) f32, count (n,) i32."""
    import torch
    assert head.is_cuda and head.dtype == torch.float32 and head.dim() == 4 and head.shape[3] == 6
    head = head.contiguous()
    n, g = head.shape[0], head.shape[1]
    k = int(num_cands)
    dev = head.device
    boxes = torch.empty((n, k, 4), dtype=torch.int32, device=dev)
    cell = torch.empty((n, k), dtype=torch.int32, device=dev)
    obj = torch.empty((n, k), dtype=torch.float32, device=dev)
    score = torch.empty((n, k), dtype=torch.float32, device=dev)
    count = torch.empty((n,), dtype=torch.int32, device=dev)
    rc = lib().fv_decode_nms(ctx.handle, ptr(head), n, g, int(image_size), float(conf_th), float(iou_th), k,
                             ptr(boxes), ptr(cell), ptr(obj), ptr(score), ptr(count))
    ctx.check(rc, 'fv_decode_nms')
    return dict(boxes=boxes, cell=cell, obj=obj, score=score, count=count)


def to_boundboxes(res, img=0):
    """Device result of decode_nms -> list[BoundBox] for one image, reference order (ascending
    score).  Coordinates are np.int64 and scores np.float32 as in the reference."""
    c = int(res['count'][img])
    b = res['boxes'][img, :c].cpu().numpy().astype(np.int64)
    o = res['obj'][img, :c].cpu().numpy()
    s = res['score'][img, :c].cpu().numpy()
    return [BoundBox(b[k, 0], b[k, 1], b[k, 2], b[k, 3], objness=o[k], classes=[s[k]]) for k in range(c)]


def letterbox_device(ctx, raw_u8, image_size, out=None):
    """uint8 HxWx3 (numpy or CUDA tensor) -> (float32 CUDA tensor (S,S,3), geometry tuple
    (h, w, pad_t, pad_b, pad_l, pad_r)) -- the device form of data.letterbox
    (reference face_detection.py:112-147)."""
    import ctypes
    import torch
    t = raw_u8 if torch.is_tensor(raw_u8) else torch.from_numpy(np.array(raw_u8, dtype=np.uint8, copy=True))
    assert t.dtype == torch.uint8 and t.dim() == 3 and t.shape[2] == 3
    t = t.cuda().contiguous()
    h, w = int(t.shape[0]), int(t.shape[1])
    S = int(image_size)
    if out is None:
        out = torch.empty((S, S, 3), dtype=torch.float32, device=t.device)
    geom = (ctypes.c_int32 * 6)()
    ctx.check(lib().fv_letterbox(ctx.handle, ptr(t), h, w, S, ptr(out), geom), 'fv_letterbox')
    return out, (h, w, geom[2], geom[3], geom[4], geom[5])


class PinnedRing(object):
    """A few pinned host buffers reused round-robin.  Page-locking a fresh 35-100 MB buffer costs 50-160 ms on this stack whenever
    torch's pinned-memory cache holds no block of that size (tools/eval_load_probe.py: batches differ in size, so every second or
    third batch missed) -- more than decoding the batch -- so the loaders write into these instead.  take() hands out the next
    buffer (grown with 25 % slack when too small) once the H2D copy that last read it has completed; the consumer calls
    copied(buffer) right after it enqueued that copy."""

    def __init__(self, n=3):
        self._pins = [None] * n
        self._events = [None] * n
        self._next = 0

    def take(self, nbytes):
        """-> uint8 pinned tensor of nbytes (a view of the slot's buffer)."""
        import torch
        i = self._next
        self._next = (i + 1) % len(self._pins)
        if self._events[i] is not None:
            self._events[i].synchronize()
            self._events[i] = None
        if self._pins[i] is None or self._pins[i].numel() < nbytes:
            self._pins[i] = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8).pin_memory()
        return self._pins[i][:nbytes]

    def untake(self):
        """The buffer of the last take() will not be used after all: the next take() hands out the same slot again."""
        self._next = (self._next - 1) % len(self._pins)

    def copied(self, buf):
        """An event on the current stream guards the slot `buf` (any view of it) came from."""
        import torch
        p = buf.data_ptr()
        for i, b in enumerate(self._pins):
            if b is not None and b.data_ptr() <= p < b.data_ptr() + max(b.numel(), 1):
                ev = torch.cuda.Event()
                ev.record()
                self._events[i] = ev
                return


def pack_images(raws, pin=True, ring=None):
    """Host side of fv_letterbox_batch: the decoded uint8 images back to back in ONE (pinned) host
    buffer -> (uint8 tensor, offsets int64 list, hw list).  ring: a PinnedRing to take the buffer from."""
    import torch
    sizes = [int(r.shape[0]) * int(r.shape[1]) * 3 for r in raws]
    if ring is not None and pin and torch.cuda.is_available():
        buf = ring.take(sum(sizes))
    else:
        buf = torch.empty(sum(sizes), dtype=torch.uint8)
        if pin and torch.cuda.is_available():
            buf = buf.pin_memory()
    view = buf.numpy()
    offs, hw, o = [], [], 0
    for r, n in zip(raws, sizes):
        a = np.asarray(r, dtype=np.uint8)
        assert a.ndim == 3 and a.shape[2] == 3
        view[o:o + n] = a.reshape(-1)
        offs.append(o); hw += [int(a.shape[0]), int(a.shape[1])]
        o += n
    return buf, offs, hw


def letterbox_batch_device(ctx, raws, image_size, device, out=None, packed=None):
    """list of uint8 HxWx3 arrays -> ((n,S,S,3) float32 CUDA tensor, [geometry tuples]) in ONE launch
    (fv_letterbox_batch); `packed` = the result of pack_images when a loader thread prepared it."""
    import ctypes
    import torch
    if packed is not None and isinstance(packed[0], str) and packed[0] == 'jpeg':
        # ('jpeg', int16 host tensor of quantised coefficients, jpeg.BatchPlan): the images are reconstructed on the device
        # (fv_jpeg_reconstruct_batch) straight into the packed RGB buffer this launch reads -- no RGB image on the host
        from . import jpeg
        _tag, coefs, plan = packed
        dbuf = jpeg.reconstruct_batch(ctx, plan, coefs.to(device, non_blocking=True), device)
        offs, hw = plan.rgb_off, plan.hw
    else:
        buf, offs, hw = packed if packed is not None else pack_images(raws)
        dbuf = buf.to(device, non_blocking=True)
    n = len(offs)
    S = int(image_size)
    if out is None:
        out = torch.empty((n, S, S, 3), dtype=torch.float32, device=device)
    geom = (ctypes.c_int32 * (6 * n))()
    rc = lib().fv_letterbox_batch(ctx.handle, ptr(dbuf), (ctypes.c_int64 * n)(*offs), (ctypes.c_int32 * (2 * n))(*hw), n, S, ptr(out), geom)
    ctx.check(rc, 'fv_letterbox_batch')
    geoms = [(hw[2 * i], hw[2 * i + 1], geom[6 * i + 2], geom[6 * i + 3], geom[6 * i + 4], geom[6 * i + 5]) for i in range(n)]
    return out, geoms

"""Host-side data path of FaceDetector: letterbox geometry, ground-truth tensor encoding and the
UCCS-format training sequence (reference face_detection.py:75-310), plus a synthetic UCCS-shaped
dataset generator (the real UCCS set is download-only and unavailable offline).

This is CPU code in the reference too (Keras `Sequence` workers); it is not part of the device
hot path.  FaceDetector itself letterboxes on the device (`fv_letterbox` via
`TrainingSequence.get_raw` / `postproc.letterbox_device`); `letterbox()` / `__getitem__` below
keep the reference's `Sequence[i] -> ({'input1': ...}, {'output': ...})` contract for callers
that want host arrays.  Pixel resampling uses a bicubic (a = -0.75, OpenCV INTER_CUBIC's kernel)
restatement; its pixel values are "parity unpinned" against cv2 (absent here) -- geometry is
pinned exactly."""
import os

import numpy as np

CSV_COLUMNS = ['FACE_ID', 'FILE', 'SUBJECT_ID', 'FACE_X', 'FACE_Y', 'FACE_WIDTH', 'FACE_HEIGHT']


def letterbox_geometry(h, w, image_size):
    """-> (w_p, h_p, pad_t, pad_b, pad_l, pad_r); odd padding puts the extra row/col at the
    bottom/right (face_detection.py:120-147)."""
    pad_t = pad_b = pad_l = pad_r = 0
    if w >= h:
        w_p, h_p = image_size, int(h / w * image_size)
        pad = image_size - h_p
        pad_t, pad_b = pad // 2, pad - pad // 2
    else:
        h_p, w_p = image_size, int(w / h * image_size)
        pad = image_size - w_p
        pad_l, pad_r = pad // 2, pad - pad // 2
    return w_p, h_p, pad_t, pad_b, pad_l, pad_r


def encode_gt(faces, h, w, image_size=416, grid=13, channels=6):
    """Ground-truth tensor of one image (face_detection.py:150-202).

    faces: array-like (n,4) of FACE_X, FACE_Y, FACE_WIDTH, FACE_HEIGHT in csv order; rows with
    any value <= 0 are skipped; later rows overwrite earlier ones in the same cell."""
    cell = image_size // grid
    _, _, pad_t, _, pad_l, _ = letterbox_geometry(h, w, image_size)
    gt = np.zeros((grid, grid, channels), np.float64)
    m = w if w >= h else h
    ox, oy = (0, pad_t) if w >= h else (pad_l, 0)
    for fx, fy, fw, fh in np.asarray(faces, dtype=np.float64).reshape(-1, 4):
        if not (fx > 0 and fy > 0 and fw > 0 and fh > 0):
            continue
        x1, y1 = int(fx), int(fy)
        x2, y2 = x1 + int(fw) - 1, y1 + int(fh) - 1
        x1p, x2p = int(x1 / m * image_size) + ox, int(x2 / m * image_size) + ox
        y1p, y2p = int(y1 / m * image_size) + oy, int(y2 / m * image_size) + oy
        xc, yc = (x1p + x2p) // 2, (y1p + y2p) // 2
        cx, cy = xc // cell, yc // cell
        gt[cy, cx, :6] = [1.0, (xc - cx * cell) / cell, (yc - cy * cell) / cell,
                          (x2 - x1 + 1) / m, (y2 - y1 + 1) / m, 1.0]
    return gt


# ----------------------------------------------------------------------------- three-scale targets
YOLO_ANCHORS = [[116, 90, 156, 198, 373, 326], [30, 61, 62, 45, 59, 119], [10, 13, 16, 30, 33, 23]]   # yolov3_detect.py:560
# The reference's decode_netout (yolov3_detect.py:354-362) SKIPS (116,90), (373,326), (62,45), (10,13), (33,23): only these
# (scale, anchor) pairs ever reach its box list, so they are the ones a face may be assigned to by default.
YOLO_ANCHORS_DECODED = ((0, 1), (1, 0), (1, 2), (2, 1))


def encode_gt_three_scale(faces, h, w, image_size=416, nclass=1, anchors=YOLO_ANCHORS, anchor_keep=YOLO_ANCHORS_DECODED):
    """Ground truth of one image for the three-scale head: [t13, t26, t52], t_s of shape (g_s, g_s, 3*(5+nclass)) float64 with
    g_s = image_size/32 * 2^s, laid out [cell][anchor][tx, ty, tw, th, objectness, classes...] like the network output.

    The single-scale encoder (face_detection.py:150-202) generalised: the same skip rule (all of X, Y, W, H > 0), the same
    integer letterbox arithmetic for the corners and the centre (`int()` truncation, `//2`), cell = centre // cell_size and
    offset = remainder / cell_size per scale.  New here (the reference never trains this head): the face goes to the ONE
    (scale, anchor) whose anchor box has the best IoU with the face box (sizes only, network pixels; ties -> the first in
    `anchor_keep` order) among the anchors the reference's decode keeps, and the box is stored in the parametrisation that
    decode_netout (yolov3_detect.py:335-387) inverts: sigma(tx) = offset, anchor_w * exp(tw) = box width in network pixels.
    Offsets are clamped to [0.5/cell, 1 - 0.5/cell] (logit of 0 is -inf; half a pixel is below the decode's int() grain).
    Objectness 1 and class 0 = 1 at the assigned slot; later rows overwrite earlier ones in the same slot."""
    S = int(image_size)
    C = 5 + nclass
    out = [np.zeros((S // 32 << s, S // 32 << s, 3 * C), np.float64) for s in range(3)]
    _, _, pad_t, _, pad_l, _ = letterbox_geometry(h, w, S)
    m = w if w >= h else h
    ox, oy = (0, pad_t) if w >= h else (pad_l, 0)
    for fx, fy, fw, fh in np.asarray(faces, dtype=np.float64).reshape(-1, 4):
        if not (fx > 0 and fy > 0 and fw > 0 and fh > 0):
            continue
        x1, y1 = int(fx), int(fy)
        x2, y2 = x1 + int(fw) - 1, y1 + int(fh) - 1
        x1p, x2p = int(x1 / m * S) + ox, int(x2 / m * S) + ox
        y1p, y2p = int(y1 / m * S) + oy, int(y2 / m * S) + oy
        xc, yc = (x1p + x2p) // 2, (y1p + y2p) // 2
        bw, bh = (x2 - x1 + 1) / m * S, (y2 - y1 + 1) / m * S          # the single-scale targets bw, bh times the network size
        best, best_iou = None, -1.0
        for (s, b) in anchor_keep:
            aw, ah = anchors[s][2 * b], anchors[s][2 * b + 1]
            inter = min(bw, aw) * min(bh, ah)
            iou = inter / (bw * bh + aw * ah - inter)
            if iou > best_iou:
                best, best_iou = (s, b), iou
        s, b = best
        g = S // 32 << s
        cell = S // g
        cx, cy = xc // cell, yc // cell
        lo, hi = 0.5 / cell, 1.0 - 0.5 / cell
        px = min(max((xc - cx * cell) / cell, lo), hi); py = min(max((yc - cy * cell) / cell, lo), hi)
        t = np.zeros(C, np.float64)
        t[0] = np.log(px / (1.0 - px)); t[1] = np.log(py / (1.0 - py))
        t[2] = np.log(bw / anchors[s][2 * b]); t[3] = np.log(bh / anchors[s][2 * b + 1])
        t[4] = 1.0; t[5] = 1.0
        out[s][cy, cx, b * C:(b + 1) * C] = t
    return out


# ----------------------------------------------------------------------------- bicubic letterbox
def _cubic_weights(t, a=-0.75):
    t = np.asarray(t, np.float64)
    w = np.empty(t.shape + (4,), np.float64)
    w[..., 0] = ((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a
    w[..., 1] = ((a + 2) * t - (a + 3)) * t * t + 1
    w[..., 2] = ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1
    w[..., 3] = 1.0 - w[..., 0] - w[..., 1] - w[..., 2]
    return w


def _resize_axis(img, n_out, axis):
    n_in = img.shape[axis]
    scale = n_in / n_out
    src = (np.arange(n_out) + 0.5) * scale - 0.5
    i0 = np.floor(src).astype(np.int64)
    wts = _cubic_weights(src - i0)
    out = 0
    for k in range(4):
        idx = np.clip(i0 - 1 + k, 0, n_in - 1)  # replicate border
        shape = [1] * img.ndim
        shape[axis] = n_out
        out = out + np.take(img, idx, axis=axis) * wts[:, k].reshape(shape)
    return out


def letterbox(image, image_size):
    """uint8/float HxWx3 -> (S,S,3) float64 in ~[0,1] + geometry (face_detection.py:112-147)."""
    img = np.asarray(image, np.float64) / 255
    h, w = img.shape[0], img.shape[1]
    w_p, h_p, pt, pb, pl, pr = letterbox_geometry(h, w, image_size)
    img = _resize_axis(_resize_axis(img, max(h_p, 1), 0), max(w_p, 1), 1)
    img = np.pad(img, ((pt, pb), (pl, pr), (0, 0)))
    return img, (h, w, pt, pb, pl, pr)


# ----------------------------------------------------------------------------- training sequence
class TrainingSequence(object):
    """Same batching contract as the reference's keras Sequence (face_detection.py:75-310):
    sorted unique FILE names, fixed consecutive slices, short last batch, hps['step'] overwritten.
    `loader(path) -> HxWx3 uint8` is injectable (PIL by default)."""

    def __init__(self, raw_data_path, hps, nn_arch, CELL_SIZE=None, cell_image_size=None, loader=None):
        import pandas as pd
        self.raw_data_path = raw_data_path
        self.hps = hps
        self.nn_arch = nn_arch
        self.gt_df = pd.read_csv(os.path.join(raw_data_path, 'training.csv'))
        self.groups = {k: v for k, v in self.gt_df.groupby('FILE')}
        self.file_names = sorted(self.groups.keys())
        self.batch_size = hps['batch_size']
        self.hps['step'] = len(self.file_names) // self.batch_size + (1 if len(self.file_names) % self.batch_size else 0)
        self.image_size = nn_arch['image_size']
        self.grid = CELL_SIZE if CELL_SIZE else self.image_size // 32
        self.loader = loader or _pil_loader
        # nn_arch['head'] == 'three_scale' (the build's extension): three target tensors per image
        self.three_scale = nn_arch.get('head', 'single') == 'three_scale'
        self.nclass = int(nn_arch.get('num_classes', 1))

    def encode(self, rows, h, w):
        """GT of one image: (G,G,6) for the reference's single-scale head, [t13, t26, t52] for the three-scale head."""
        if self.three_scale:
            return encode_gt_three_scale(rows, h, w, self.image_size, self.nclass)
        return encode_gt(rows, h, w, self.image_size, self.grid, self.nn_arch['bb_info_c_size'])

    def __len__(self):
        return self.hps['step']

    def __getitem__(self, index):
        names = self.file_names[index * self.batch_size:(index + 1) * self.batch_size]
        images, gts = [], []
        for name in names:
            raw = self.loader(os.path.join(self.raw_data_path, name))
            img, (h, w, *_rest) = letterbox(raw, self.image_size)
            df = self.groups[name]
            gts.append(self.encode(df.iloc[:, 3:7].values, h, w))
            images.append(img)
        if self.three_scale:
            return ({'input1': np.asarray(images)}, {'output%d' % s: np.asarray([g[s] for g in gts]) for s in range(3)})
        return ({'input1': np.asarray(images)}, {'output': np.asarray(gts)})

    def get_raw(self, index):
        """Same batch, but images stay raw uint8 (decoded only): the letterbox then runs on the
        device (fv_letterbox).  -> (list of HxWx3 uint8 arrays, (b,G,G,6) float32 GT tensors)."""
        names = self.file_names[index * self.batch_size:(index + 1) * self.batch_size]
        raws, gts = [], []
        for name in names:
            raw = self.loader(os.path.join(self.raw_data_path, name))
            df = self.groups[name]
            gts.append(self.encode(df.iloc[:, 3:7].values, raw.shape[0], raw.shape[1]))
            raws.append(raw)
        if self.three_scale:
            return raws, [np.asarray([g[s] for g in gts], np.float32) for s in range(3)]
        return raws, np.asarray(gts, np.float32)


def _pil_loader(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert('RGB'))


# ----------------------------------------------------------------------------- synthetic data
def synth_gt_batch(batch, image_size=416, faces_per_image=5, seed=1234, grid=None):
    """Synthetic GT tensors (B,G,G,6) float32 through encode_gt."""
    rng = np.random.default_rng(seed)
    grid = grid or image_size // 32
    out = np.zeros((batch, grid, grid, 6), np.float32)
    for b in range(batch):
        h, w = int(rng.integers(300, 1100)), int(rng.integers(300, 1100))
        n = max(1, int(rng.poisson(faces_per_image)))
        fw = rng.uniform(12, w / 4, n); fh = rng.uniform(12, h / 4, n)
        fx = rng.uniform(1, w - fw - 1); fy = rng.uniform(1, h - fh - 1)
        out[b] = encode_gt(np.stack([fx, fy, fw, fh], 1), h, w, image_size, grid)
    return out


def make_synthetic_uccs(root, n_images=4, seed=0, csv_name='training.csv', sizes=None):
    """Write a tiny UCCS-format dataset: JPEGs of uniform noise + csv with the reference's column
    order.  Returns the DataFrame."""
    import pandas as pd
    from PIL import Image
    rng = np.random.default_rng(seed)
    os.makedirs(root, exist_ok=True)
    sizes = sizes or [(480, 640), (640, 480), (600, 800), (416, 416)]
    rows = []
    fid = 0
    for k in range(n_images):
        h, w = sizes[k % len(sizes)]
        name = 'synth_%04d.jpg' % k
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(root, name), quality=90)
        for _ in range(int(rng.integers(1, 4))):
            fw = float(rng.uniform(20, w / 3)); fh = float(rng.uniform(20, h / 3))
            fx = float(rng.uniform(1, w - fw - 1)); fy = float(rng.uniform(1, h - fh - 1))
            rows.append([fid, name, int(rng.integers(1, 100)), round(fx, 1), round(fy, 1), round(fw, 1), round(fh, 1)])
            fid += 1
    df = pd.DataFrame(rows, columns=CSV_COLUMNS)
    df.to_csv(os.path.join(root, csv_name), index=False)
    return df

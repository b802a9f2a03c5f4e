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

    def __len__(self):
        return self.hps['step']

    def __getitem__(self, index):
        names = self.file_names[index * self.batch_size:(index + 1) * self.batch_size]
        images, gts = [], []
        for name in names:
            raw = self.loader(os.path.join(self.raw_data_path, name))
            img, (h, w, *_rest) = letterbox(raw, self.image_size)
            df = self.groups[name]
            gts.append(encode_gt(df.iloc[:, 3:7].values, h, w, self.image_size, self.grid, self.nn_arch['bb_info_c_size']))
            images.append(img)
        return ({'input1': np.asarray(images)}, {'output': np.asarray(gts)})

    def get_raw(self, index):
        """Same batch, but images stay raw uint8 (decoded only): the letterbox then runs on the
        device (fv_letterbox).  -> (list of HxWx3 uint8 arrays, (b,G,G,6) float32 GT tensors)."""
        names = self.file_names[index * self.batch_size:(index + 1) * self.batch_size]
        raws, gts = [], []
        for name in names:
            raw = self.loader(os.path.join(self.raw_data_path, name))
            df = self.groups[name]
            gts.append(encode_gt(df.iloc[:, 3:7].values, raw.shape[0], raw.shape[1], self.image_size, self.grid,
                                 self.nn_arch['bb_info_c_size']))
            raws.append(raw)
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

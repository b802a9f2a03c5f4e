"""ORACLE -- test infrastructure only.

Plain-loop restatements of the reference's host-side (CPU, integer/float64) pieces of the
FaceDetector path.  Pinned by tests/golden/{gt_encoder,weight_reader,decode_netout}.npz.
Paths are relative to /root/reference/src/space.
"""
import math
import struct

import numpy as np


# ----------------------------------------------------------------------------- letterbox geometry
def letterbox_geometry(h, w, image_size):
    """face_detection.py:120-147 -> (w_p, h_p, pad_t, pad_b, pad_l, pad_r)."""
    pad_t = pad_b = pad_l = pad_r = 0
    if w >= h:
        w_p = image_size
        h_p = int(h / w * image_size)
        pad = image_size - h_p
        pad_t = pad // 2
        pad_b = pad // 2 + (pad % 2)
    else:
        h_p = image_size
        w_p = int(w / h * image_size)
        pad = image_size - w_p
        pad_l = pad // 2
        pad_r = pad // 2 + (pad % 2)
    return w_p, h_p, pad_t, pad_b, pad_l, pad_r


def letterbox_pixels(raw_u8, image_size):
    """Plain-loop float64 restatement of image/255 -> bicubic resize (a=-0.75, half-pixel centres,
    replicated border = OpenCV INTER_CUBIC's definition) -> zero pad (face_detection.py:112-147).
    PARITY UNPINNED against cv2 (not installed): used to check the product's CPU and device
    letterbox against one independent statement of the same formula."""
    raw = np.asarray(raw_u8, np.float64) / 255
    h, w = raw.shape[:2]
    S = image_size
    w_p, h_p, pt, pb, pl, pr = letterbox_geometry(h, w, S)

    def wts(t):
        a = -0.75
        w0 = ((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a
        w1 = ((a + 2) * t - (a + 3)) * t * t + 1
        w2 = ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1
        return [w0, w1, w2, 1 - w0 - w1 - w2]

    out = np.zeros((S, S, 3), np.float64)
    xs = []
    for xi in range(w_p):
        fx = (xi + 0.5) * (w / w_p) - 0.5
        sx = math.floor(fx)
        xs.append(([min(max(sx - 1 + i, 0), w - 1) for i in range(4)], wts(fx - sx)))
    for yi in range(h_p):
        fy = (yi + 0.5) * (h / h_p) - 0.5
        sy = math.floor(fy)
        wy = wts(fy - sy)
        rows = [raw[min(max(sy - 1 + j, 0), h - 1)] for j in range(4)]
        for xi, (ix, wx) in enumerate(xs):
            acc = 0.0
            for j in range(4):
                acc = acc + wy[j] * sum(wx[i] * rows[j][ix[i]] for i in range(4))
            out[pt + yi, pl + xi] = acc
    return out


# ----------------------------------------------------------------------------- GT encoder
def gt_encode_image(rows, h, w, image_size=416, grid=13, channels=6):
    """face_detection.py:150-202 for one image.

    rows: iterable of (FACE_X, FACE_Y, FACE_WIDTH, FACE_HEIGHT) in csv order.
    Returns (grid, grid, channels) float64."""
    cell = image_size // grid
    _, _, pad_t, _, pad_l, _ = letterbox_geometry(h, w, image_size)
    gt = np.zeros((grid, grid, channels), np.float64)
    for fx, fy, fw, fh in rows:
        if not (fx > 0 and fy > 0 and fw > 0 and fh > 0):  # fd.py:154-156
            continue
        x1 = int(fx); y1 = int(fy)
        x2 = x1 + int(fw) - 1; y2 = y1 + int(fh) - 1
        wb = x2 - x1 + 1; hb = y2 - y1 + 1
        if w >= h:  # fd.py:168-177
            x1_p = int(x1 / w * image_size); y1_p = int(y1 / w * image_size) + pad_t
            x2_p = int(x2 / w * image_size); y2_p = int(y2 / w * image_size) + pad_t
        else:
            x1_p = int(x1 / h * image_size) + pad_l; y1_p = int(y1 / h * image_size)
            x2_p = int(x2 / h * image_size) + pad_l; y2_p = int(y2 / h * image_size)
        xc = (x1_p + x2_p) // 2; yc = (y1_p + y2_p) // 2
        cx = xc // cell; cy = yc // cell
        bx = (xc - cx * cell) / cell; by = (yc - cy * cell) / cell
        m = w if w >= h else h
        gt[cy, cx, 0] = 1.0
        gt[cy, cx, 1] = bx; gt[cy, cx, 2] = by
        gt[cy, cx, 3] = wb / m; gt[cy, cx, 4] = hb / m
        gt[cy, cx, 5] = 1.0
    return gt


def gt_encode_three_scale(rows, h, w, image_size=416, nclass=1, allowed=((0, 1), (1, 0), (1, 2), (2, 1))):
    """The build's three-scale target encoder restated (face_vijnana_yolov3_amd/data.py:encode_gt_three_scale; the reference
    trains no three-scale head, so there is nothing of the reference's to pin this to except its decode: the round trip
    decode_netout(encode(box)) below, yolov3_detect.py:335-404).  Box / centre integers as face_detection.py:150-177; anchors
    yolov3_detect.py:560; `allowed` = the (scale, anchor) pairs the reference's decode does not skip (yolov3_detect.py:354-362)."""
    anchors = ((116, 90, 156, 198, 373, 326), (30, 61, 62, 45, 59, 119), (10, 13, 16, 30, 33, 23))
    S = image_size
    nch = 5 + nclass
    grids = [S // 32, S // 16, S // 8]
    out = [np.zeros((g, g, 3, nch), np.float64) for g in grids]
    _, _, pad_t, _, pad_l, _ = letterbox_geometry(h, w, S)
    for fx, fy, fw, fh in rows:
        if min(fx, fy, fw, fh) <= 0:
            continue
        x1 = int(fx); y1 = int(fy)
        x2 = x1 + int(fw) - 1; y2 = y1 + int(fh) - 1
        if w >= h:
            x1_p = int(x1 / w * S); y1_p = int(y1 / w * S) + pad_t
            x2_p = int(x2 / w * S); y2_p = int(y2 / w * S) + pad_t
            bw = (x2 - x1 + 1) / w * S; bh = (y2 - y1 + 1) / w * S
        else:
            x1_p = int(x1 / h * S) + pad_l; y1_p = int(y1 / h * S)
            x2_p = int(x2 / h * S) + pad_l; y2_p = int(y2 / h * S)
            bw = (x2 - x1 + 1) / h * S; bh = (y2 - y1 + 1) / h * S
        xc = (x1_p + x2_p) // 2; yc = (y1_p + y2_p) // 2
        ious = []
        for (sc, b) in allowed:
            aw = anchors[sc][2 * b]; ah = anchors[sc][2 * b + 1]
            i = min(aw, bw) * min(ah, bh)
            ious.append(i / (aw * ah + bw * bh - i))
        sc, b = allowed[int(np.argmax(ious))]          # argmax returns the first maximum
        cell = S // grids[sc]
        col = xc // cell; row = yc // cell
        ox = (xc % cell) / cell; oy = (yc % cell) / cell
        half = 0.5 / cell
        ox = half if ox < half else (1 - half if ox > 1 - half else ox)
        oy = half if oy < half else (1 - half if oy > 1 - half else oy)
        v = out[sc][row, col, b]
        v[:] = 0.0
        v[0] = math.log(ox) - math.log1p(-ox); v[1] = math.log(oy) - math.log1p(-oy)
        v[2] = math.log(bw / anchors[sc][2 * b]); v[3] = math.log(bh / anchors[sc][2 * b + 1])
        v[4] = 1.0; v[5] = 1.0
    return [o.reshape(o.shape[0], o.shape[1], 3 * nch) for o in out]


def training_batches(file_names, batch_size):
    """face_detection.py:84-90, 103-104, 207: sorted unique file names, fixed consecutive
    slices, short last batch.  Returns list of lists of file names."""
    step = len(file_names) // batch_size + (1 if len(file_names) % batch_size else 0)
    return [file_names[i * batch_size:(i + 1) * batch_size] for i in range(step)]


# ----------------------------------------------------------------------------- Darknet .weights
def darknet_header_len(buf):
    """yolov3_detect.py:70-77: 3 int32 then 8 more bytes iff major*10+minor >= 2 (both < 1000)."""
    major, minor, _rev = struct.unpack_from('iii', buf, 0)
    return 12 + (8 if (major * 10 + minor) >= 2 and major < 1000 and minor < 1000 else 4)


def read_darknet_weights(buf, layers):
    """yolov3_detect.py:90-121.

    layers: list of (index, keras_kernel_shape (kh,kw,cin,cout), has_bn) in ascending index
    order (only the layers that exist in the model).  Returns ({'conv_i': [kernel(,bias)],
    'bnorm_i': [gamma,beta,mean,var]}, floats_consumed)."""
    data = np.frombuffer(buf, dtype='<f4', offset=darknet_header_len(buf))
    off = 0
    out = {}

    def take(n):
        nonlocal off
        off += n
        return data[off - n:off]

    for idx, shape, has_bn in layers:
        cout = shape[3]
        if has_bn:
            beta = take(cout); gamma = take(cout); mean = take(cout); var = take(cout)
            out['bnorm_%d' % idx] = [gamma, beta, mean, var]
        bias = None if has_bn else take(cout)
        kern = take(int(np.prod(shape))).reshape(tuple(reversed(shape))).transpose(2, 3, 1, 0)
        out['conv_%d' % idx] = [kern] if has_bn else [kern, bias]
    return out, off


# ----------------------------------------------------------------------------- secondary: 3-scale decode
def _sig64(x):
    return 1.0 / (1.0 + math.exp(-x))


def decode_netout(netout, anchors, anchor_idx, obj_thresh, net_h, net_w):
    """yolov3_detect.py:335-387 (anchor skip list yd.py:354-362).  float32 sigmoid on the
    array as the reference does (np.exp on float32); returns rows
    [xmin,ymin,xmax,ymax,objness,classes...] (relative units)."""
    gh, gw = netout.shape[:2]
    no = netout.reshape(gh, gw, 3, -1).astype(np.float32).copy()
    sig = lambda a: (np.float32(1.) / (np.float32(1.) + np.exp(-a))).astype(np.float32)
    no[..., :2] = sig(no[..., :2]); no[..., 4:] = sig(no[..., 4:])
    keep = {0: (1,), 1: (0, 2), 2: (1,)}[anchor_idx]
    rows = []
    for i in range(gh * gw):
        r, c = i // gw, i % gw
        for b in keep:
            conf = no[r, c, b, 4]
            if conf < obj_thresh:
                continue
            x, y, w, h = no[r, c, b, :4]
            x = (c + x) / gw; y = (r + y) / gh
            w = anchors[2 * b] * np.exp(w) / net_w; h = anchors[2 * b + 1] * np.exp(h) / net_h
            rows.append([x - w / 2, y - h / 2, x + w / 2, y + h / 2, conf] + list(no[r, c, b, 5:]))
    return rows


def correct_yolo_boxes(rows, image_h, image_w, net_h, net_w):
    """yolov3_detect.py:389-404 (note the reference's `new_h = net_w` in the else branch)."""
    if (float(net_w) / image_w) < (float(net_h) / image_h):
        new_w = net_w; new_h = (image_h * net_w) / image_w
    else:
        new_h = net_w; new_w = (image_w * net_h) / image_h
    x_off, x_sc = (net_w - new_w) / 2. / net_w, float(new_w) / net_w
    y_off, y_sc = (net_h - new_h) / 2. / net_h, float(new_h) / net_h
    for r in rows:
        r[0] = int((r[0] - x_off) / x_sc * image_w); r[2] = int((r[2] - x_off) / x_sc * image_w)
        r[1] = int((r[1] - y_off) / y_sc * image_h); r[3] = int((r[3] - y_off) / y_sc * image_h)


def _overlap(x1, x2, x3, x4):
    if x3 < x1:
        return 0 if x4 < x1 else min(x2, x4) - x1
    return 0 if x2 < x3 else min(x2, x4) - x3


def do_nms(rows, nms_thresh):
    """yolov3_detect.py:426-444: per-class greedy NMS, suppression = class prob := 0."""
    if not rows:
        return
    ncls = len(rows[0]) - 5
    for c in range(ncls):
        order = np.argsort([-r[5 + c] for r in rows], kind='stable')
        for a in range(len(order)):
            ia = order[a]
            if rows[ia][5 + c] == 0:
                continue
            A = rows[ia]
            for b in range(a + 1, len(order)):
                B = rows[order[b]]
                inter = _overlap(A[0], A[2], B[0], B[2]) * _overlap(A[1], A[3], B[1], B[3])
                uni = (A[2] - A[0]) * (A[3] - A[1]) + (B[2] - B[0]) * (B[3] - B[1]) - inter
                if float(inter) / uni >= nms_thresh:
                    B[5 + c] = 0

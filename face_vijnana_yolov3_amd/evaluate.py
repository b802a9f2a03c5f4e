"""Detection accuracy of FaceDetector.evaluate's output: the reference's `cal_mAP_fd`
(evaluate.py:27-127) and the IoU-threshold sweep of its `main` (evaluate.py:337-355).

Host code, as in the reference (pandas/NumPy/SciPy on a few thousand boxes); not part of the device
hot path.  What it restates, line by line:

* solution csv (no header): FILE, x, y, w, h, confidence  -- what `FaceDetector.evaluate` writes
  (face_detection.py:733-737); ground truth csv (header): FACE_ID, FILE, SUBJECT_ID, FACE_X,
  FACE_Y, FACE_WIDTH, FACE_HEIGHT.  Boxes are (x, y, x + w, y + h) (evaluate.py:50-53, 61-64).
* per image: IoU of every (gt, detection) pair with `bbox_iou` (yd.py:165-194; pairs with IoU <= 0
  are dropped, evaluate.py:67), greedy one-to-one assignment in descending IoU order
  (evaluate.py:81-95); a detection's IoU is that of its assigned gt, -1 if it got none;
* images without detections, and images where no (gt, detection) pair overlaps, are skipped -- their
  detections never enter the result, false positives included (evaluate.py:43-47, 77); neither do
  detections on images that are not in the ground truth (the loop runs over ground-truth images);
* detections sorted by confidence, descending (evaluate.py:104); precision / recall after each one
  with TP = "IoU >= iou_th" and the recall denominator = ALL ground-truth rows (evaluate.py:108-119);
* mAP = integral of the linear interpolant precision(recall) from rs[0] to rs[-1]
  (`interp1d` + `quad`, evaluate.py:124-125) -- the same SciPy calls are made here.

Parity unpinned: the reference function cannot be executed -- `sol_df.iat[:, 6] = -1.0`
(evaluate.py:31, 36) raises "iAt based indexing can only have integer indexers" in every pandas
release, so no golden vectors can be minted; `bbox_iou` itself is pinned (tests/golden/iou_cases.npz).
Two places where the committed text is ill-defined are resolved as follows: (1) the initial IoU of an
unmatched detection is the -1.0 those two lines try to assign; (2) `res_df` is assigned at loop index
k == 0 only (evaluate.py:97-100), which raises NameError when the first ground-truth image has no
detections -- here results are simply collected over all images.  Ties in IoU / confidence keep file
order (stable sorts; the reference's `sort_values` default is not stable)."""
import numpy as np


def _interval_overlap(a1, a2, b1, b2):
    """yd.py:165-178 with its branch structure (the result may be negative for touching boxes)."""
    if b1 < a1:
        return 0 if b2 < a1 else min(a2, b2) - a1
    return 0 if a2 < b1 else min(a2, b2) - b1


def bbox_iou_xyxy(b1, b2):
    """yd.py:183-194 on (xmin, ymin, xmax, ymax) tuples; ZeroDivisionError/nan behaviour not reproduced:
    a zero union gives nan."""
    iw = _interval_overlap(b1[0], b1[2], b2[0], b2[2])
    ih = _interval_overlap(b1[1], b1[3], b2[1], b2[3])
    inter = iw * ih
    union = (b1[2] - b1[0]) * (b1[3] - b1[1]) + (b2[2] - b2[0]) * (b2[3] - b2[1]) - inter
    return float(inter) / union if union != 0 else float('nan')


def iou_matrix_device(ctx, gt_boxes, det_boxes):
    """All (gt, detection) IoUs of one or many images in ONE kernel launch (fv_bbox_iou_pairs): gt_boxes (n,4),
    det_boxes (m,4) as x, y, w, h -> (n, m) float64, bit-identical to bbox_iou_xyxy on the same pairs."""
    import torch
    from ._lib import lib, ptr
    g = np.asarray(gt_boxes, np.float64).reshape(-1, 4); d = np.asarray(det_boxes, np.float64).reshape(-1, 4)
    n, m = len(g), len(d)
    if n == 0 or m == 0:
        return np.zeros((n, m))
    gx = np.stack([g[:, 0], g[:, 1], g[:, 0] + g[:, 2], g[:, 1] + g[:, 3]], 1)
    dx = np.stack([d[:, 0], d[:, 1], d[:, 0] + d[:, 2], d[:, 1] + d[:, 3]], 1)
    dev = torch.device('cuda', ctx.device)
    a = torch.from_numpy(np.repeat(gx, m, axis=0)).to(dev); b = torch.from_numpy(np.tile(dx, (n, 1))).to(dev)
    out = torch.empty(n * m, dtype=torch.float64, device=dev)
    ctx.check(lib().fv_bbox_iou_pairs(ctx.handle, ptr(a), ptr(b), n * m, ptr(out)), 'fv_bbox_iou_pairs')
    return out.cpu().numpy().reshape(n, m)


def match_image(gt_boxes, det_boxes, ctx=None):
    """Greedy assignment of one image (evaluate.py:46-95).  gt_boxes (n,4), det_boxes (m,4) as
    x, y, w, h.  Returns the (m,) IoU assigned to each detection (-1 = unmatched), or None when no pair
    overlaps (the reference then drops the image's detections, false positives included).  ctx: an fv
    Context -> the pair IoUs come from the device kernel (same values)."""
    out = np.full(len(det_boxes), -1.0)
    pairs = []
    if ctx is not None:
        with np.errstate(invalid='ignore'):
            m = iou_matrix_device(ctx, gt_boxes, det_boxes)
            pairs = [(int(i), int(j), float(m[i, j])) for i, j in zip(*np.nonzero(m > 0.))]
    for i, g in enumerate(gt_boxes if ctx is None else []):
        gb = (g[0], g[1], g[0] + g[2], g[1] + g[3])
        for j, d in enumerate(det_boxes):
            iou = bbox_iou_xyxy(gb, (d[0], d[1], d[0] + d[2], d[1] + d[3]))
            if iou > 0.:
                pairs.append((i, j, iou))
    if not pairs:
        return None                           # evaluate.py:77: the image then contributes nothing at all
    pairs.sort(key=lambda t: -t[2])          # stable: ties keep (gt, detection) order
    used_g, used_d = set(), set()
    for i, j, iou in pairs:
        if i in used_g or j in used_d:
            continue
        out[j] = iou
        used_g.add(i); used_d.add(j)
    return out


def detection_ious(gt_df, sol_df, ctx=None):
    """-> (confidences, assigned IoUs) of every detection on a ground-truth image that has detections,
    in ground-truth image order (evaluate.py:39-101)."""
    sol_groups = {k: v for k, v in sol_df.groupby(0, sort=True)}
    conf, ious = [], []
    for image_id, df in gt_df.groupby('FILE', sort=True):
        rel = sol_groups.get(image_id)
        if rel is None or len(rel) == 0:
            continue
        iou = match_image(df.iloc[:, 3:7].to_numpy(dtype=np.float64), rel.iloc[:, 1:5].to_numpy(dtype=np.float64), ctx)
        if iou is None:
            continue
        conf.append(rel.iloc[:, 5].to_numpy(dtype=np.float64))
        ious.append(iou)
    if not conf:
        return np.zeros(0), np.zeros(0)
    return np.concatenate(conf), np.concatenate(ious)


def pr_curve(conf, ious, gt_count, iou_th):
    """evaluate.py:104-122."""
    order = np.argsort(-conf, kind='stable')
    tp = np.cumsum(ious[order] >= iou_th)
    n = np.arange(1, len(order) + 1)
    return tp / n, tp / float(gt_count)


def integrate_pr(ps, rs):
    """evaluate.py:124-125: quad over the linear interpolant of (rs, ps)."""
    from scipy.integrate import quad
    from scipy.interpolate import interp1d
    if len(rs) < 2 or rs[0] == rs[-1]:
        return 0.0
    func = interp1d(rs, ps)
    return quad(lambda x: func(x), rs[0], rs[-1])[0]


def cal_mAP_fd(gt_path, sol_path, iou_th, ctx=None):
    """-> (ps, rs, mAP) exactly as the reference's signature (evaluate.py:27, 127); ctx: optional fv Context
    (pair IoUs on the device)."""
    import pandas as pd
    sol_df = pd.read_csv(sol_path, header=None)
    gt_df = pd.read_csv(gt_path)
    conf, ious = detection_ious(gt_df, sol_df, ctx)
    ps, rs = pr_curve(conf, ious, gt_df.shape[0], iou_th)
    return ps, rs, integrate_pr(ps, rs)


def cal_mAP_sweep(gt_path, sol_path, iou_ths=None):
    """The sweep of evaluate.py:337-347 (IoU 0.50 ... 0.95): -> list of (iou_th, mAP) and their mean
    (the README's "mAP" is the mean of AP50..AP95).  The matching does not depend on the threshold and
    is done once."""
    import pandas as pd
    iou_ths = np.arange(0.5, 1.0, 0.05) if iou_ths is None else iou_ths
    sol_df = pd.read_csv(sol_path, header=None)
    gt_df = pd.read_csv(gt_path)
    conf, ious = detection_ious(gt_df, sol_df)
    res = []
    for th in iou_ths:
        ps, rs = pr_curve(conf, ious, gt_df.shape[0], th)
        res.append((float(th), integrate_pr(ps, rs)))
    return res, float(np.mean([m for _, m in res])) if res else 0.0


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description='mAP of a FaceDetector.evaluate solution file (reference evaluate.py cal_map_fd)')
    ap.add_argument('--mode', default='cal_map_fd')
    ap.add_argument('--gt_path', required=True)
    ap.add_argument('--sol_path', required=True)
    a = ap.parse_args(argv)
    if a.mode != 'cal_map_fd':
        raise SystemExit('only cal_map_fd is part of the FaceDetector path (SURVEY 8f)')
    res, mean = cal_mAP_sweep(a.gt_path, a.sol_path)
    for th, m in res:
        print('{0:1.2f}'.format(th), m)
    print('mean', mean)


if __name__ == '__main__':
    main()

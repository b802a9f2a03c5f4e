"""TEST INFRASTRUCTURE ONLY (imported by tests/ alone).

Loop-level restatement of the reference's cal_mAP_fd (evaluate.py:27-127) with plain Python
containers, following its control flow statement by statement: per ground-truth image, the table
of (gt, detection, IoU) triples with IoU > 0 (evaluate.py:46-71), sorted by IoU descending
(evaluate.py:79), then the `while` loop that takes the head triple, writes its IoU to the
detection and deletes every triple sharing its gt or its detection (evaluate.py:82-95); the
precision / recall walk over the detections in confidence order (evaluate.py:104-122) and the
interp1d + quad integral (evaluate.py:124-125).

PARITY UNPINNED: the reference function raises at evaluate.py:31 under every pandas release
(`.iat[:, 6] = -1.0`), so it cannot be run to mint vectors; IoU itself is pinned through
oracle/postproc.bbox_iou against tests/golden/iou_cases.npz (integer boxes)."""
import csv

from .host_oracle import _overlap


def _iou(A, B):
    """yolov3_detect.py:183-194 on float boxes (xmin, ymin, xmax, ymax); the integer form of the same
    function is pinned by tests/golden/iou_cases.npz through oracle/postproc.bbox_iou."""
    inter = _overlap(A[0], A[2], B[0], B[2]) * _overlap(A[1], A[3], B[1], B[3])
    uni = (A[2] - A[0]) * (A[3] - A[1]) + (B[2] - B[0]) * (B[3] - B[1]) - inter
    return float(inter) / uni if uni != 0 else float('nan')


def cal_mAP_fd(gt_path, sol_path, iou_th):
    sol = {}
    with open(sol_path) as f:
        for row in csv.reader(f):
            if row:
                sol.setdefault(row[0], []).append([float(v) for v in row[1:6]] + [-1.0])
    gt = {}
    gt_count = 0
    with open(gt_path) as f:
        rd = csv.reader(f)
        next(rd)
        for row in rd:
            if row:
                gt.setdefault(row[1], []).append([float(v) for v in row[3:7]])
                gt_count += 1
    res = []
    for image_id in sorted(gt.keys()):
        if image_id not in sol:
            continue
        dets = sol[image_id]
        triples = []
        for i, g in enumerate(gt[image_id]):
            gb = (g[0], g[1], g[0] + g[2], g[1] + g[3])
            for j, d in enumerate(dets):
                iou = _iou(gb, (d[0], d[1], d[0] + d[2], d[1] + d[3]))
                if iou > 0.:
                    triples.append((i, j, iou))
        if not triples:
            continue          # evaluate.py:77: such an image contributes no detections at all
        triples.sort(key=lambda t: -t[2])
        while triples:
            i, j, iou = triples[0]
            dets[j][5] = iou
            triples = [t for t in triples if t[0] != i]
            triples = [t for t in triples if t[1] != j]
        res.extend(dets)
    res.sort(key=lambda d: -d[4])
    ps, rs = [], []
    tp = 0
    for n, d in enumerate(res, 1):
        if d[5] >= iou_th:
            tp += 1
        ps.append(tp / n)
        rs.append(tp / gt_count)
    if len(rs) < 2 or rs[0] == rs[-1]:
        return ps, rs, 0.0
    from scipy.integrate import quad
    from scipy.interpolate import interp1d
    func = interp1d(rs, ps)
    return ps, rs, quad(lambda x: func(x), rs[0], rs[-1])[0]

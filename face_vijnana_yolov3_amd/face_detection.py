"""FaceDetector: drop-in for the train / evaluate / test / detect surface of the reference's
`src/space/face_detection.py` (class FaceDetector fd.py:66-949, main() fd.py:951-985), running the
network and the detect post-processing on MI355X through the C ABI.

Same constructor (`conf = json['fd_conf']`), same methods, same config keys, same side-effect
files (solution csv with 6 columns and no header, <test_path>/results/*_detected.jpg,
ratios.csv, the model files).  Differences, all documented in DESIGN.md:
  * model files are real HDF5 in Keras' WEIGHT layout (hdf5_lite.py, a pure-Python reader/writer: no HDF5 library in the
    main interpreter): the reference's `model.load_weights` / h5py read them and the reference's face_detector.h5 /
    yolov3_base.h5 load here; there is no `model_config`, so Keras' `load_model` does not take them
  * the grid is image_size/32 (the reference hard-codes 13, consistent only at 416)
  * multi_gpu=True means one process per GPU with RCCL all-reduce instead of keras.utils.multi_gpu_model towers: main()
    starts num_gpus ranks itself (parallel.launch_ranks -> torch.distributed.run); a FaceDetector constructed directly in a
    process without WORLD_SIZE trains on one GPU
  * evaluate() tolerates images without ground-truth rows and a missing arial.ttf
  * evaluate()/test() read ahead and run the network on batches of hps['eval_batch_size'] images (default 48, default_eval_batch(); the
    reference's loop is batch 1, fd.py:632-883) -- same rows in the same order (text-identical for the same head output; the
    network's float32 summation order depends on the batch size, so scores may differ in the 7th digit between batch sizes)
  * train() and test() decode baseline JPEGs in two halves (jpeg.py): Huffman decoding on host threads, dequantisation / IDCT /
    chroma upsampling / colour conversion on the device, bit-identical to Pillow's pixels (hps['device_jpeg'] = false: Pillow);
    evaluate() draws on the image and keeps decoding it with Pillow
  * nn_arch['head'] = 'three_scale' (default 'single' = the reference's 13x13x6 head) selects the full three-scale YOLOv3
    graph (yolov3_detect.py:217-311) with nn_arch['num_classes'] (default 1) classes: trained with the build's
    objectness / box / class loss on targets from data.encode_gt_three_scale, detected through the reference's
    decode_netout / correct_yolo_boxes / do_nms chain (fv_yolo_decode_nms)
"""
import glob
import json
import os
import platform
import time
from concurrent.futures import ThreadPoolExecutor, wait

import numpy as np

from . import data
from .postproc import BoundBox, PinnedRing, decode_nms, letterbox_batch_device, pack_images, to_boundboxes

DEBUG = True


def effective_cpus():
    """CPUs this process may really use: the affinity mask, cut by the cgroup CPU quota where one is set (a 1-GPU share of an
    MI355X host shows 256 logical CPUs and a quota of 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 8)
    try:                                        # cgroup v2
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        try:                                    # cgroup v1
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0 and per > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def default_loader_threads():
    """hps['loader_threads'] when the configuration does not set it: the CPUs the process may use, at least 4, at most 32 (the
    reference asks for 4 or 8 Keras workers, fd.py:621-627).  Measured (bench.py `test_loop`, eval batch 32, a 16-CPU share of a
    256-thread host): 8 / 16 / 32 / 64 threads -> 1990 / 2073 / 2058 / 2021 img/s -- flat beyond the quota; the loop is bound
    by the device there (forward 0.40 ms/img = 2500 img/s, plus the JPEG reconstruction and letterbox kernels beside it)."""
    return max(4, min(32, effective_cpus()))


def default_eval_batch(image_size):
    """hps['eval_batch_size'] when the configuration does not set it: 48 images per forward where the first layer's output fits
    one 2 GiB buffer descriptor (batch x image_size^2 x 32 floats <= 2^29 elements: up to 416 and beyond), else the largest
    multiple of 8 that does (608: 40).  48, not 32: the loop is bound by the device (bench.py `test_loop`), and the 128-row tile
    rounds of the 26^2 / 52^2 layers come out whole at multiples of 24 images (device only, network + decode/NMS at 416^2:
    batch 24 / 32 / 40 / 48 / 64 = 2519 / 2360 / 2510 / 2610 / 2565 img/s, tools/eval_batch_probe.py)."""
    from .engine import Engine
    fit = Engine.max_infer_batch(image_size)
    return 48 if fit >= 48 else max(1, fit // 8 * 8 if fit >= 8 else fit)


def map_all(pool, fn, items):
    """pool.map that returns only when EVERY task has finished, also when one raised (Executor.map stops waiting at the first
    exception and leaves the running tasks writing into the buffer they were given); re-raises the first exception."""
    futs = [pool.submit(fn, it) for it in items]
    wait(futs)
    return [f.result() for f in futs]


class FaceDetector(object):
    """Face detector using the Darknet-53 base of YOLOv3."""

    MODEL_PATH = 'face_detector.h5'
    BASE_MODEL_PATH = 'yolov3_base.h5'
    DARKNET_WEIGHTS_PATH = 'yolov3.weights'
    OUTPUT_FILE_NAME = 'solution.csv'
    EVALUATION_FILE_NAME = 'eval.csv'
    CELL_SIZE = 13

    TrainingSequence = data.TrainingSequence

    def __init__(self, conf, device=None):
        from .engine import Engine
        self.conf = conf
        self.raw_data_path = conf['raw_data_path']
        self.hps = conf['hps']
        self.nn_arch = conf['nn_arch']
        self.model_loading = conf['model_loading']
        self.image_size = int(self.nn_arch['image_size'])
        if self.image_size % 32:
            raise ValueError('image_size must be a multiple of 32 (network stride)')
        if int(self.nn_arch.get('bb_info_c_size', 6)) != 6:
            raise ValueError('bb_info_c_size must be 6')
        self.grid = self.image_size // 32
        self.cell_image_size = self.image_size // self.grid
        self.rank = int(os.environ.get('RANK', 0))
        self.world = int(os.environ.get('WORLD_SIZE', 1)) if conf.get('multi_gpu') else 1
        if device is None:           # FV_DEVICE: several ranks on ONE device, to rehearse N > 1 on a one-GPU box (gloo transport)
            device = int(os.environ.get('FV_DEVICE', os.environ.get('LOCAL_RANK', 0)))
        self.three_scale = self.nn_arch.get('head', 'single') == 'three_scale'
        if self.nn_arch.get('head', 'single') not in ('single', 'three_scale'):
            raise ValueError("nn_arch.head must be 'single' or 'three_scale'")
        if self.three_scale:
            from .yolov3 import Yolov3
            self.nclass = int(self.nn_arch.get('num_classes', 1))
            self.model = Yolov3(device, out_channels=3 * (5 + self.nclass))
        else:
            self.model = Engine(device)
        # BN moving statistics as the reference's stack updates them (Keras 2.2.4: zero-debiased); "bn_zero_debias": false in the
        # configuration selects the plain EMA
        self.model.bn_zero_debias = bool(conf.get('bn_zero_debias', True))
        if self.model_loading:
            self.model.load(self.MODEL_PATH)
        elif self.three_scale:
            self._load_base_three_scale()
        else:
            self._load_base()
            self._init_head()

    # ------------------------------------------------------------------ model construction
    def _load_base(self):
        """YOLOV3Base (fd.py:384-600): cached base file, else Darknet weights, else -- because
        neither can be downloaded offline -- synthetic initialisation (announced)."""
        from . import weights
        eng = self.model
        if self.conf.get('yolov3_base_model_load') and os.path.exists(self.BASE_MODEL_PATH):
            eng.load(self.BASE_MODEL_PATH, require_all=False)        # the base file holds no head (fd.py:393-396)
        elif os.path.exists(self.DARKNET_WEIGHTS_PATH):
            p, s = weights.read_darknet_base(self.DARKNET_WEIGHTS_PATH, eng.layers, eng.n_params, eng.n_state)
            eng.set_params(p, s)
            if self.rank == 0:                                       # base.save('yolov3_base.h5') (fd.py:596-598)
                weights.write_keras_h5(self.BASE_MODEL_PATH, eng.layers[:-1], p, s, nested=None)
        else:
            print('FaceDetector: neither %s nor %s found; using synthetic base weights'
                  % (self.BASE_MODEL_PATH, self.DARKNET_WEIGHTS_PATH))
            eng.init_synthetic(seed=7)

    def _load_base_three_scale(self):
        """Three-scale model: layers 75..105 random-init (BN layers ~ N(0, 2/fan_in), detection convs glorot-uniform, zero
        bias), the 52 base layers from the cached base file / the Darknet file when present (same flat layout as Engine)."""
        from . import weights
        m = self.model
        m.init_synthetic(seed=7)
        base = m.layers[:52]
        n_p = base[-1]['beta_off'] + base[-1]['cout']; n_s = base[-1]['var_off'] + base[-1]['cout']
        if self.conf.get('yolov3_base_model_load') and os.path.exists(self.BASE_MODEL_PATH):
            from .hdf5_lite import is_hdf5, read_hdf5
            if is_hdf5(self.BASE_MODEL_PATH):
                p, st, _found = weights.from_keras_datasets(read_hdf5(self.BASE_MODEL_PATH)[0], base, n_p, n_s)
                m.load_base(p, st)
            else:
                with open(self.BASE_MODEL_PATH, 'rb') as f:
                    d = np.load(f)
                    m.load_base(d['params'][:n_p], d['state'][:n_s])
        elif os.path.exists(self.DARKNET_WEIGHTS_PATH):
            p, st = weights.read_darknet_base(self.DARKNET_WEIGHTS_PATH, base, n_p, n_s)
            m.load_base(p, st)
        else:
            print('FaceDetector: neither %s nor %s found; using synthetic base weights'
                  % (self.BASE_MODEL_PATH, self.DARKNET_WEIGHTS_PATH))

    @property
    def YOLOV3Base(self):
        """The Darknet-53 base as a model of its own (fd.py:384-600: "partial yolo3 model from the input layer to the add_23
        layer"; the reference builds its detector around it, fd.py:344-352, and FaceIdentifier reuses it,
        face_identification.py:323).  Here: a view of THIS detector's base weights with `predict` (numpy, as Keras),
        `predict_device` (CUDA tensor) and `save` (the yolov3_base.h5 layout, fd.py:598) -- fv_forward_base underneath.
        Single-scale head: a live view of the detector's engine.  Three-scale head: a SNAPSHOT -- the base weights are copied into a
        cached Engine at every access of the property (read it again after training)."""
        if self.three_scale:
            # the three-scale model keeps its base in the same flat layout at the same offsets: lend it to an Engine
            from .engine import Engine
            eng = getattr(self, '_base_engine', None)
            if eng is None:
                eng = self._base_engine = Engine(self.model.ctx.device)     # one Engine, reused: only the weights are refreshed per access
            nb = eng.layers[-2]
            n_p, n_s = nb['beta_off'] + nb['cout'], nb['var_off'] + nb['cout']
            eng.params[:n_p].copy_(self.model.params[:n_p]); eng.state[:n_s].copy_(self.model.state[:n_s])
            return YoloV3BaseModel(eng)
        return YoloV3BaseModel(self.model)

    def _init_head(self, seed=None):
        """Keras default for the 'output' Conv2D: glorot_uniform kernel, zero bias (fd.py:348-352)."""
        import torch
        eng = self.model
        d = eng.layers[-1]
        n = d['cout'] * 9 * d['cin']
        lim = float(np.sqrt(6.0 / (9 * d['cin'] + 9 * d['cout'])))
        g = torch.Generator().manual_seed(0 if seed is None else seed)
        eng.params[d['w_off']:d['w_off'] + n] = ((torch.rand(n, generator=g) * 2 - 1) * lim).to(eng.dev)
        eng.params[d['beta_off']:d['beta_off'] + d['cout']] = 0

    # ------------------------------------------------------------------ train (fd.py:602-630)
    def train(self):
        from .parallel import DataParallelTrainer
        seq = self.TrainingSequence(self.raw_data_path, self.hps, self.nn_arch, self.grid, self.cell_image_size)
        trainer = DataParallelTrainer(self.model, world_size=self.world, rank=self.rank)
        hp = self.hps
        steps = len(seq)
        rng = np.random.default_rng(0)
        feeder = BatchFeeder(seq, self.world, self.rank, int(hp.get('loader_threads', default_loader_threads())))
        log_every = max(1, int(hp.get('log_every', 1)))
        for epoch in range(hp['epochs']):
            order = rng.permutation(steps)  # Keras fit_generator shuffles batch order (shuffle=True)
            if self.rank == 0:
                print('Epoch %d/%d' % (epoch + 1, hp['epochs']))
            def after_step(k, loss, item):
                if item is None:
                    if self.rank == 0:
                        print('%d/%d - skipped (fewer images than ranks)' % (k + 1, steps))
                    return
                # Keras prints the loss of the MERGED batch at every step (verbose=1, fd.py:621-627); reading it is a host
                # sync (and one small all-reduce when world > 1): hps['log_every'] = n prints every n-th step only
                if ((k + 1) % log_every == 0 and DEBUG) or k + 1 == steps:
                    lv = trainer.merged_loss(loss, item[2])          # collective: every rank calls it
                    if self.rank == 0:
                        print('%d/%d - loss: %.4f' % (k + 1, steps, lv))
            run_pipelined(self.model, trainer, feeder, [int(i) for i in order], self.image_size, hp, after_step)
        feeder.close()
        if self.rank == 0:
            print('Save the model.')
            self.model.save(self.MODEL_PATH)
        trainer.shutdown()

    # ------------------------------------------------------------------ detect (fd.py:885-949)
    def detect(self, image):
        """image: (1,S,S,3) array in [0,1] -> list[BoundBox], ascending score, at most num_cands."""
        return self.detect_batch(image)[0]

    def detect_batch(self, images):
        """(B,S,S,3) array or CUDA tensor -> one list[BoundBox] per image (what evaluate()/test() run per batch)."""
        return self._detect_collect(self._detect_launch(images))

    def _detect_launch(self, images):
        """Queue network + decode/NMS/top-k for a batch and the copy of the (small) result into pinned host memory; no host sync.
        evaluate()/test() queue batch k+1 before they read batch k (_detect_collect), so the device never waits for the host's
        box bookkeeping."""
        import torch
        x = images if torch.is_tensor(images) else np.asarray(images, dtype=np.float32)
        if self.three_scale:
            # the reference's decode_netout -> correct_yolo_boxes -> do_nms chain (yolov3_detect.py:335-444; its driver loops over the
            # images, yd.py:596-604) on the letterboxed images: ONE launch pair for the batch (fv_yolo_decode_nms_batch), results
            # into pinned memory behind an event, like the single-scale head
            from .yolov3 import decode_nms_batch
            S = self.image_size
            ys = self.model.predict_device(x)
            res = decode_nms_batch(self.model.ctx, ys[0], ys[1], ys[2], (S, S), (S, S), obj_thresh=self.hps['face_conf_th'],
                                   nms_thresh=self.hps['nms_iou_th'])
            host = {k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True).copy_(v, non_blocking=True) for k, v in res.items()}
            ev = torch.cuda.Event()
            ev.record()
            return ('three_scale', host, ev, int(ys[0].shape[0]))
        y = self.model.predict_device(x)
        res = decode_nms(self.model.ctx, y, self.image_size, self.hps['face_conf_th'], self.hps['nms_iou_th'],
                         self.hps['num_cands'])
        # one D2H copy per tensor for the whole batch, ordered behind THIS batch's kernels only
        host = {k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True).copy_(v, non_blocking=True) for k, v in res.items()}
        ev = torch.cuda.Event()
        ev.record()
        return ('single', host, ev, int(y.shape[0]))

    def _detect_collect(self, launched):
        tag, host, ev, n = launched
        ev.synchronize()
        if tag == 'three_scale':
            return [self._three_scale_boxes(host, b) for b in range(n)]
        return [to_boundboxes(host, b) for b in range(n)]

    def _three_scale_boxes(self, host, b):
        """Image b of a decoded batch -> detect()'s own tail (fd.py:942-947): boxes in network pixels, score > 0, ascending score, at
        most num_cands.  The selection runs on the arrays (BoundBox.get_score: the probability of the arg-max class, capped at 1);
        only the <= num_cands survivors become BoundBox objects -- a frame can hold thousands of candidates."""
        n = int(host['count'][b])
        bx = host['boxes'][b, :n].numpy().astype(np.int64); ob = host['objness'][b, :n].numpy(); cl = host['classes'][b, :n].numpy()
        if n == 0:
            return []
        score = np.minimum(cl[np.arange(n), np.argmax(cl, axis=1)], np.float32(1.0))
        keep = np.nonzero(score > 0)[0]
        keep = keep[np.argsort(score[keep], kind='stable')][:self.hps['num_cands']]
        return [BoundBox(bx[k, 0], bx[k, 1], bx[k, 2], bx[k, 3], objness=ob[k], classes=list(cl[k])) for k in keep]

    # ------------------------------------------------------------------ evaluate / test
    def _project_back(self, boxes, geom):
        """Undo the letterbox (fd.py:700-710): per coordinate `np.min([v * w / S, w])` / `np.min([np.max([v - pad, 0]) * w / S, h])`.
        Evaluated for all boxes of an image at once: the same IEEE operations in the same order on float64 (int -> float64, subtract,
        max, multiply, divide, min), so every value -- and the text str() makes of it in the csv row -- is the reference's; the
        per-box form costs eight np.min / np.max calls on two-element lists per box (~1.5 ms for an image with 60 boxes: with the
        three-scale head that, not the GPU, set the rate of test(): 709 img/s at batch 16, 537 at 32)."""
        if not boxes:
            return
        h, w, pad_t, _pb, pad_l, _pr = geom
        S = self.image_size
        c = np.array([[b.xmin, b.ymin, b.xmax, b.ymax] for b in boxes], dtype=np.float64)
        if w >= h:
            xs = np.minimum(c[:, (0, 2)] * w / S, w)
            ys = np.minimum(np.maximum(c[:, (1, 3)] - pad_t, 0) * w / S, h)
        else:
            xs = np.minimum(np.maximum(c[:, (0, 2)] - pad_l, 0) * h / S, w)
            ys = np.minimum(c[:, (1, 3)] * h / S, h)
        for i, b in enumerate(boxes):
            b.xmin, b.xmax = xs[i, 0], xs[i, 1]
            b.ymin, b.ymax = ys[i, 0], ys[i, 1]

    @staticmethod
    def _write_rows(f, file_name, boxes):
        base = os.path.basename(file_name)
        for b in boxes[:60]:   # the writers hard-code 60 (fd.py:729, 870)
            f.write(base + ',' + str(b.xmin) + ',' + str(b.ymin) + ',' + str(b.xmax - b.xmin) + ','
                    + str(b.ymax - b.ymin) + ',' + str(b.get_score()) + '\n')

    def _files(self, test_path):
        from .parallel import shard_files
        names = sorted(glob.glob(os.path.join(test_path, '*.jpg')))
        return shard_files(names, self.world, self.rank) if self.world > 1 else names

    def _detect_files(self, files, need_raw=True):
        """Yield (file_name, raw image, boxes in image coordinates) in file order.  The reference's evaluate()/test() loops
        (fd.py:632-883) decode, letterbox and predict one image at a time; here a thread pool decodes batch k+1 (PIL releases
        the GIL) while batch k is letterboxed in one launch (fv_letterbox_batch) and runs ONE forward + ONE decode/NMS launch
        -- batch 1 is the slowest operating point of the network (1.3 ms/img against 0.4 at batch 16+).  Two deep: the loader
        thread fills a reused pinned buffer (PinnedRing) with batch k+1 while batch k is in the device queue and the host turns
        batch k-1's result into BoundBoxes and rows."""
        bs = max(1, int(self.hps.get('eval_batch_size', default_eval_batch(self.image_size))))
        chunks = [files[i:i + bs] for i in range(0, len(files), bs)]
        if not chunks:
            return
        threads = max(1, min(bs, int(self.hps.get('loader_threads', default_loader_threads()))))
        use_jpeg = not need_raw and bool(self.hps.get('device_jpeg', True))
        if getattr(self, '_eval_ring', None) is None:
            self._eval_ring = PinnedRing(3)          # kept across evaluate()/test() calls: page-locking is the expensive part
        ring = self._eval_ring
        with ThreadPoolExecutor(max_workers=threads) as pool, ThreadPoolExecutor(max_workers=1) as one:
            def load(chunk):
                if use_jpeg:       # test() never looks at the pixels on the host: Huffman-decode only, the rest on the device
                    import torch
                    from . import jpeg
                    datas = list(pool.map(lambda f: open(f, 'rb').read(), chunk))
                    infos = [jpeg.parse(d) for d in datas]
                    if all(i is not None for i in infos):
                        plan = jpeg.BatchPlan(infos)
                        buf = ring.take(2 * plan.total_coefs).view(torch.int16)     # decoded straight into a reused pinned buffer
                        view = buf.numpy()
                        try:
                            map_all(pool, lambda i: jpeg.entropy_decode(datas[i], infos[i], view[plan.coef_off[i]:plan.coef_off[i] + int(infos[i].total_coefs)]),
                                    range(len(chunk)))
                            return None, ('jpeg', buf, plan)
                        except ValueError:
                            # damaged scan data (a bad Huffman code, a run past the block): libjpeg -- the reference's reader
                            # (skimage.io.imread, fd.py:798) -- only warns and returns an image; so the batch goes through Pillow.
                            # map_all has waited for every decoder task: nothing writes into the slot any more
                            ring.untake()
                raws = list(pool.map(data._pil_loader, chunk))
                return raws, pack_images(raws, ring=ring)
            def finish(done):
                chunk_, raws_, geoms_, launched = done
                for i, (name, boxes, geom) in enumerate(zip(chunk_, self._detect_collect(launched), geoms_)):
                    self._project_back(boxes, geom)
                    yield name, (raws_[i] if raws_ is not None else None), boxes
            # The device half of the input path (H2D copy of the batch -- 70 MB of JPEG coefficients at 32 images: ~3 ms of PCIe time --,
            # fv_jpeg_reconstruct_batch, fv_letterbox_batch) runs on its OWN stream: batch k+1 is staged while batch k's forward computes
            # (on one stream the copy sat in front of every forward: 16.7 ms per batch of 32 where the forward takes 12.8).
            import torch
            dev = self.model.dev
            main = torch.cuda.current_stream(dev)
            if getattr(self, '_stage_stream', None) is None:
                self._stage_stream = torch.cuda.Stream(device=dev)
            side = self._stage_stream

            def stage(loaded):
                raws, packed = loaded
                with torch.cuda.stream(side):
                    self.model.ctx.set_stream(side.cuda_stream)
                    try:
                        x, geoms = letterbox_batch_device(self.model.ctx, raws, self.image_size, dev, packed=packed)
                    finally:
                        self.model.ctx.set_stream(main.cuda_stream)
                    ring.copied(packed[1] if isinstance(packed[0], str) else packed[0])   # event on the staging stream: the pinned slot is free after it
                    ev = torch.cuda.Event()
                    ev.record(side)
                return raws, geoms, x, ev

            prev = None
            pending = one.submit(load, chunks[0])
            staged = stage(pending.result())
            if len(chunks) > 1:
                pending = one.submit(load, chunks[1])
            for k, chunk in enumerate(chunks):
                raws, geoms, x, ev = staged
                main.wait_event(ev)
                x.record_stream(main)              # allocated on the staging stream, consumed on the compute stream
                cur = (chunk, raws, geoms, self._detect_launch(x))
                if k + 1 < len(chunks):            # the host waits for the decode of k+1 while the GPU runs batch k, then stages it beside it
                    loaded = pending.result()
                    if k + 2 < len(chunks):
                        pending = one.submit(load, chunks[k + 2])
                    staged = stage(loaded)
                if prev is not None:               # read batch k-1 now that batch k is in the queue
                    for item in finish(prev):
                        yield item
                prev = cur
            for item in finish(prev):
                yield item

    def evaluate(self):
        import pandas as pd
        test_path = self.conf['test_path']
        out_path = self.conf['output_file_path']
        res_dir = os.path.join(test_path, 'results')
        from .parallel import ensure_process_group, merge_rank_files, part_path, reset_dir_before_shards
        if self.world > 1:
            ensure_process_group(self.model.dev)
        reset_dir_before_shards(res_dir, self.rank)     # rank 0 empties it BEFORE any rank writes into it
        gt_df = pd.read_csv(os.path.join(test_path, 'validation.csv'))
        groups = {k: v for k, v in gt_df.groupby('FILE')}
        files = self._files(test_path)
        ratios = []
        with open(part_path(out_path, self.world, self.rank), 'w') as f:
            for n, (file_name, raw, boxes) in enumerate(self._detect_files(files)):
                if DEBUG:
                    print(n + 1, '/', len(files), file_name)
                self._write_rows(f, file_name, boxes)
                if len(boxes) == 0:
                    continue
                base = os.path.basename(file_name)
                gt_boxes = []
                df = groups.get(base)
                if df is not None:
                    for i in range(df.shape[0]):
                        v = df.iloc[i, 3:7].values.astype(np.float64)
                        if not np.all(v > 0):
                            continue
                        xmin, ymin = int(v[0]), int(v[1])
                        xmax, ymax = int(xmin + v[2] - 1), int(ymin + v[3] - 1)
                        gt_boxes.append(BoundBox(xmin, ymin, xmax, ymax, objness=1., classes=[1.0]))
                        if ymax != ymin:
                            ratios.append((xmax - xmin) / (ymax - ymin))
                img = draw_boxes(raw, gt_boxes, self.hps['face_conf_th'], (255, 0, 0))
                img = draw_boxes(img, boxes, self.hps['face_conf_th'], (0, 255, 0))
                new_name = base[:-4] + '_detected' + base[-4:]
                print(new_name)
                from PIL import Image
                Image.fromarray(img.astype('uint8')).save(os.path.join(res_dir, new_name))
        # one 6-column solution csv and one ratios.csv, whatever the number of ranks (cal_mAP_fd reads the former)
        merge_rank_files(out_path, self.world, self.rank)
        if self.world == 1:
            pd.DataFrame({'ratio': ratios}).to_csv('ratios.csv')          # fd.py:780-781
        else:
            pd.DataFrame({'ratio': ratios}).to_csv(part_path('ratios.csv', self.world, self.rank), index=False)
            merge_rank_files('ratios.csv', self.world, self.rank, header_lines=1)
            if self.rank == 0:
                pd.read_csv('ratios.csv').to_csv('ratios.csv')          # the reference's layout: running index column

    def test(self):
        test_path = self.conf['test_path']
        out_path = self.conf['output_file_path']
        files = self._files(test_path)
        from .parallel import ensure_process_group, merge_rank_files, part_path
        if self.world > 1:
            ensure_process_group(self.model.dev)
        with open(part_path(out_path, self.world, self.rank), 'w') as f:
            for n, (file_name, _raw, boxes) in enumerate(self._detect_files(files, need_raw=False)):
                if DEBUG:
                    print(n + 1, '/', len(files), file_name)
                self._write_rows(f, file_name, boxes)
        merge_rank_files(out_path, self.world, self.rank)


class YoloV3BaseModel(object):
    """What FaceDetector.YOLOV3Base hands out: input (B,S,S,3) in [0,1] -> (B,S/32,S/32,1024), the output of the last
    residual add of the Darknet-53 base (fd.py:384-600).  A view of the engine it was given (FaceDetector.YOLOV3Base says which)."""
    trainable = True                                   # fd.py:396, 597

    def __init__(self, engine):
        self._eng = engine
        self.output_channels = int(engine.layers[-1]['cin'])

    def predict_device(self, x):
        return self._eng.predict_base_device(x)

    def predict(self, x):
        return self.predict_device(x).cpu().numpy()

    def save(self, path):
        """base.save('yolov3_base.h5') (fd.py:598): one HDF5 group per Keras layer, no head."""
        from . import weights
        eng = self._eng
        weights.write_keras_h5(path, eng.layers[:-1], eng.params.cpu().numpy(), eng.state.cpu().numpy(), nested=None)


class BatchFeeder(object):
    """Loader side of train() -- the role of the reference's Keras Sequence workers (fd.py:75-310,
    fit_generator(workers=4|8) fd.py:621-627): for batch `index`, this rank's tower slice is JPEG-decoded
    by a thread pool (PIL releases the GIL), its GT tensors are encoded, and the decoded images are
    packed back to back into one pinned host buffer.  Resize + pad happen on the device, in one launch
    per batch (fv_letterbox_batch).  One batch is prepared ahead while the previous step runs."""

    def __init__(self, seq, world, rank, threads=8):
        self.seq, self.world, self.rank = seq, world, rank
        self.pool = ThreadPoolExecutor(max_workers=max(1, threads))
        self.one = ThreadPoolExecutor(max_workers=1)
        self.device_jpeg = bool(seq.hps.get('device_jpeg', True))    # hps.device_jpeg = false: decode with Pillow on the host
        self.pending = None
        # three pinned staging buffers, reused round-robin; a buffer is rewritten only after the H2D copy that read it has
        # completed (event recorded by the consumer through copied())
        self.ring = PinnedRing(3)

    def _load_jpeg(self, seq, names, pin):
        """The batch as quantised JPEG coefficients (jpeg.py): the files are read and Huffman-decoded by the pool straight into
        one pinned int16 buffer; dequantisation, IDCT, chroma upsampling and colour conversion happen on the device.  None when a
        file is not a baseline JPEG this decoder takes (the batch then goes through Pillow)."""
        import torch
        from . import jpeg
        datas = list(self.pool.map(lambda nm: open(os.path.join(seq.raw_data_path, nm), 'rb').read(), names))
        infos = [jpeg.parse(d) for d in datas]
        if any(i is None for i in infos):
            return None
        plan = jpeg.BatchPlan(infos)
        slot = None
        if pin:
            raw = slot = self.ring.take(2 * plan.total_coefs)
            buf = raw.view(torch.int16)
        else:
            buf = torch.empty(plan.total_coefs, dtype=torch.int16)
        view = buf.numpy()
        try:
            map_all(self.pool, lambda i: jpeg.entropy_decode(datas[i], infos[i], view[plan.coef_off[i]:plan.coef_off[i] + int(infos[i].total_coefs)]),
                    range(len(names)))
        except ValueError:
            # damaged scan data: libjpeg (the reference's reader, fd.py:112) warns and still returns an image -- Pillow path
            # (every decoder task has finished: map_all waits for all of them before it raises)
            if slot is not None:
                self.ring.untake()
            return None
        return ('jpeg', buf, plan), [(i.height, i.width) for i in infos], slot

    def copied(self, slot):
        """Called by the consumer right after it enqueued the H2D copy of the buffer `slot` (on the stream that copies)."""
        self.ring.copied(slot)

    def load(self, index):
        import torch
        from .parallel import slice_batch
        seq = self.seq
        names = seq.file_names[index * seq.batch_size:(index + 1) * seq.batch_size]
        sl = slice_batch(len(names), self.world, self.rank) if self.world > 1 else (0, len(names), 1.0)
        if sl is None:
            return None            # fewer images than ranks: skipped on every rank alike
        lo, hi, weight = sl
        mine = names[lo:hi]
        pin = torch.cuda.is_available()
        jp = self._load_jpeg(seq, mine, pin) if (seq.loader is data._pil_loader and self.device_jpeg) else None
        if jp is not None:
            packed, shapes, slot = jp
        elif seq.loader is data._pil_loader:
            # sizes from the JPEG headers first, then every worker decodes its image straight into its
            # slice of the one pinned buffer (no second pass over ~100 MB per batch on one thread)
            from PIL import Image
            ims = [Image.open(os.path.join(seq.raw_data_path, nm)) for nm in mine]
            hw, offs, o = [], [], 0
            for im in ims:
                w, h = im.size
                hw += [h, w]; offs.append(o); o += h * w * 3
            slot = None
            if pin:
                buf = slot = self.ring.take(o)
            else:
                buf = torch.empty(o, dtype=torch.uint8)
            view = buf.numpy()

            def work(i):
                with ims[i] as im:
                    a = np.asarray(im.convert('RGB'))
                view[offs[i]:offs[i] + a.size] = a.reshape(-1)
            list(self.pool.map(work, range(len(ims))))
            packed = (buf, offs, hw)
            shapes = [(hw[2 * i], hw[2 * i + 1]) for i in range(len(ims))]
        else:
            slot = None
            raws = list(self.pool.map(lambda nm: seq.loader(os.path.join(seq.raw_data_path, nm)), mine))
            packed = pack_images(raws, pin=pin)
            shapes = [(r.shape[0], r.shape[1]) for r in raws]
        enc = [seq.encode(seq.groups[nm].iloc[:, 3:7].values, h, w) for nm, (h, w) in zip(mine, shapes)]
        if getattr(seq, 'three_scale', False):      # [t13, t26, t52] per image -> three stacked tensors
            yt = tuple(torch.from_numpy(np.asarray([e[s] for e in enc], np.float32)) for s in range(3))
            yt = tuple(t.pin_memory() for t in yt) if pin else yt
        else:
            yt = torch.from_numpy(np.asarray(enc, np.float32))
            yt = yt.pin_memory() if pin else yt
        return packed, yt, weight, (self, slot)

    def prefetch(self, index):
        self.pending = self.one.submit(self.load, index)

    def take(self):
        return self.pending.result()

    def close(self):
        self.one.shutdown(); self.pool.shutdown()


def train_on_item(engine, trainer, item, image_size, hp):
    """One optimisation step on what BatchFeeder.load returned, everything on the compute stream: H2D copy, device JPEG
    reconstruction / letterbox, fv_train_step (+ gradient all-reduce when world > 1), Adam.  run_pipelined() is the overlapped form."""
    packed, y, weight, (feeder, slot) = item
    x, _ = letterbox_batch_device(engine.ctx, None, image_size, engine.dev, packed=packed)
    if slot is not None:
        feeder.copied(slot)
    yd = [t.to(engine.dev, non_blocking=True) for t in y] if isinstance(y, (tuple, list)) else y.to(engine.dev, non_blocking=True)
    return trainer.train_on_batch(x, yd, hp['lr'], hp['beta_1'], hp['beta_2'], hp.get('decay', 0.0), weight=weight)


class DeviceStager(object):
    """The device half of the input path -- H2D copy of the batch (89 MB of JPEG coefficients or raw pixels at batch 40: 2-4 ms
    of PCIe time), fv_jpeg_reconstruct_batch, fv_letterbox_batch, targets -- on its OWN stream, so that batch k+1 is staged while
    step k computes instead of in front of step k+1 (measured with the loader in the loop: 56.7 -> 55.3-56.0 ms per step)."""

    def __init__(self, engine, image_size):
        import torch
        self.eng, self.S = engine, image_size
        self.stream = torch.cuda.Stream(device=engine.dev)

    def stage(self, item):
        import torch
        if item is None:
            return None
        packed, y, weight, (feeder, slot) = item
        main = torch.cuda.current_stream(self.eng.dev)
        with torch.cuda.stream(self.stream):
            self.eng.ctx.set_stream(self.stream.cuda_stream)      # the library's launches of this block go to the staging stream
            try:
                x, _ = letterbox_batch_device(self.eng.ctx, None, self.S, self.eng.dev, packed=packed)
            finally:
                self.eng.ctx.set_stream(main.cuda_stream)
            if slot is not None:
                feeder.copied(slot)                               # event on the staging stream: the pinned buffer is free after it
            yd = [t.to(self.eng.dev, non_blocking=True) for t in y] if isinstance(y, (tuple, list)) else y.to(self.eng.dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return x, yd, weight, ev

    def use(self, staged):
        """Make the compute stream wait for the staged batch; -> (x, targets, weight)."""
        import torch
        x, yd, weight, ev = staged
        main = torch.cuda.current_stream(self.eng.dev)
        main.wait_event(ev)
        for t in [x] + (list(yd) if isinstance(yd, (tuple, list)) else [yd]):
            t.record_stream(main)                                 # allocated on the staging stream, consumed on the compute stream
        return x, yd, weight


def run_pipelined(engine, trainer, feeder, indices, image_size, hp, after_step=None):
    """One optimisation step per batch index in `indices`, the input side pipelined two deep: while step k computes, the host
    threads decode batch k+2 (BatchFeeder) and the staging stream copies, reconstructs and letterboxes batch k+1 (DeviceStager).
    after_step(k, loss, item) is called one step late -- once step k+1 is in the queue -- so that reading the loss (a host sync)
    never leaves the GPU idle (item None: batch skipped on all ranks)."""
    n = len(indices)
    if n == 0:
        return
    stager = DeviceStager(engine, image_size)
    feeder.prefetch(indices[0])
    item = feeder.take()
    if n > 1:
        feeder.prefetch(indices[1])
    staged = stager.stage(item)
    lagged = None                                                 # (k, loss copy, item) of the previous step, reported one step late
    for k in range(n):
        cur, item_k, loss = staged, item, None
        if cur is not None:
            x, yd, weight = stager.use(cur)
            loss = trainer.train_on_batch(x, yd, hp['lr'], hp['beta_1'], hp['beta_2'], hp.get('decay', 0.0), weight=weight)
            loss = loss.clone()                                   # the engine reuses its loss buffer: keep this step's value
        if k + 1 < n:
            item = feeder.take()                                  # the host waits for the decode of k+1 while the GPU runs step k
            if k + 2 < n:
                feeder.prefetch(indices[k + 2])
            staged = stager.stage(item)
        # reading a loss is a host sync: report step k-1 now that step k is in the queue, so the GPU never waits for the host
        if after_step is not None and lagged is not None:
            after_step(*lagged)
        lagged = (k, loss, item_k)
    if after_step is not None and lagged is not None:
        after_step(*lagged)


def _font():
    from PIL import ImageFont
    for name in ('arial.ttf', 'DejaVuSans.ttf'):
        try:
            return ImageFont.truetype(name, 15)
        except OSError:
            continue
    return ImageFont.load_default()


def draw_boxes(image, boxes, conf_th, color):
    """Rectangle + score text for every box whose score passes conf_th (the visual contract of
    the reference's draw_boxes_v3, yolov3_detect.py:505-540)."""
    from PIL import Image, ImageDraw
    im = Image.fromarray(np.asarray(image).astype('uint8'))
    dr = ImageDraw.Draw(im)
    font = _font()
    for b in boxes:
        s = float(b.get_score())
        if s < conf_th:
            continue
        x0, y0, x1, y1 = [float(v) for v in (b.xmin, b.ymin, b.xmax, b.ymax)]
        dr.rectangle([min(x0, x1), min(y0, y1), max(x0, x1), max(y0, y1)], outline=color, width=2)
        dr.text((min(x0, x1), max(min(y0, y1) - 16, 0)), '%.3f' % s, fill=color, font=font)
    return np.asarray(im)


def main():
    """Reads ./face_vijnana_yolov3.json (Windows: _win) and dispatches on fd_conf.mode (fd.py:951-985)."""
    name = 'face_vijnana_yolov3_win.json' if platform.system() == 'Windows' else 'face_vijnana_yolov3.json'
    with open(name, 'r') as f:
        conf = json.load(f)['fd_conf']
    if conf['mode'] not in ('train', 'evaluate', 'test'):
        return
    n = int(conf.get('num_gpus', 1)) if conf.get('multi_gpu') else 1
    if n > 1 and 'WORLD_SIZE' not in os.environ:
        # multi_gpu_model(model, gpus=num_gpus) (fd.py:358-371) is one process driving num_gpus towers; here it is one process
        # per GPU, started from this one BEFORE it makes any GPU call -- the reference's command line stays what it was
        from .parallel import launch_ranks
        raise SystemExit(launch_ranks(n, ['-m', 'face_vijnana_yolov3_amd.face_detection']))
    fd = FaceDetector(conf)
    ts = time.time()
    getattr(fd, conf['mode'])()
    print('Elasped time: {0:f}s'.format(time.time() - ts))


if __name__ == '__main__':
    main()

"""Where the loader-inclusive step loses time against the resident-input step: host-side phases of run_pipelined's loop
(take = wait for the decoded batch, stage = enqueue the device half, train = issue the step's launches), for several loader
thread counts, against the plain loop's host issue time."""
import os, sys, time, tempfile

def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np, torch
    from PIL import Image
    from face_vijnana_yolov3_amd import data
    from face_vijnana_yolov3_amd.engine import Engine
    from face_vijnana_yolov3_amd.parallel import DataParallelTrainer
    from face_vijnana_yolov3_amd.face_detection import BatchFeeder, DeviceStager
    B, S, N = 40, 416, 16
    HP = dict(lr=1e-4, beta_1=0.99, beta_2=0.99, decay=0.0)
    eng = Engine(0); eng.init_synthetic(seed=7)
    tr = DataParallelTrainer(eng, world_size=1, rank=0)
    x = torch.rand((B, S, S, 3)).cuda(); y = torch.from_numpy(data.synth_gt_batch(B, S, seed=1)).cuda()
    for _ in range(5):
        tr.train_on_batch(x, y, **HP)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); issue = 0.0
    for _ in range(N):
        a = time.perf_counter(); tr.train_on_batch(x, y, **HP); issue += time.perf_counter() - a
    torch.cuda.synchronize()
    print('plain loop: %.2f ms per step, host issue %.2f ms per step' % ((time.perf_counter() - t0) / N * 1e3, issue / N * 1e3), flush=True)
    with tempfile.TemporaryDirectory() as root:
        rng = np.random.default_rng(0)
        sizes = [(768, 1024), (1024, 768), (720, 1280), (600, 800)]
        rows = []
        for k in range(2 * B):
            h, w = sizes[k % 4]
            lo = rng.integers(0, 256, (h // 16 + 1, w // 16 + 1, 3), dtype=np.uint8)
            Image.fromarray(lo).resize((w, h), Image.BICUBIC).save(os.path.join(root, 'img_%04d.jpg' % k), quality=90)
            rows.append([k, 'img_%04d.jpg' % k, 1, 10.0, 10.0, 50.0, 60.0])
        import pandas as pd
        pd.DataFrame(rows, columns=data.CSV_COLUMNS).to_csv(os.path.join(root, 'training.csv'), index=False)
        for threads, cached in ((16, False), (8, False), (4, False), (16, True)):
            seq = data.TrainingSequence(root, dict(batch_size=B, step=1), {'image_size': S, 'bb_info_c_size': 6})
            f = BatchFeeder(seq, 1, 0, threads)
            st = DeviceStager(eng, S)
            cache = [f.load(0), f.load(1)] if cached else None
            def get(k):
                return cache[k % 2] if cached else f.take()
            def pre(k):
                if not cached:
                    f.prefetch(k % 2)
            ph = dict(take=0.0, stage=0.0, train=0.0)
            for rep in range(2):          # first repetition warms up
                for k_ in ph: ph[k_] = 0.0
                pre(0); item = get(0); pre(1); staged = st.stage(item)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for k in range(N):
                    a = time.perf_counter()
                    xx, yd, w = st.use(staged)
                    tr.train_on_batch(xx, yd, weight=w, **HP)
                    b = time.perf_counter()
                    item = get(k + 1)
                    c = time.perf_counter()
                    pre(k + 2)
                    staged = st.stage(item)
                    d = time.perf_counter()
                    ph['train'] += b - a; ph['take'] += c - b; ph['stage'] += d - c
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print('threads %2d cached %-5s: %.2f ms per step; host: train %.2f, take %.2f, stage %.2f ms' %
                  (threads, cached, dt / N * 1e3, ph['train'] / N * 1e3, ph['take'] / N * 1e3, ph['stage'] / N * 1e3), flush=True)
            if not cached:
                f.take()
            f.close()


if __name__ == '__main__':
    main()

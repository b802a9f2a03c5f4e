"""Loader-only rates of BatchFeeder with the JPEG split and with Pillow, and where the split's host time goes."""
import os, sys, time, tempfile

def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np, torch
    from PIL import Image
    from face_vijnana_yolov3_amd import data, jpeg
    from face_vijnana_yolov3_amd.face_detection import BatchFeeder
    torch.cuda.init()
    B, S = 40, 416
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
        for threads in (16, 8, 32):
            for dj in (False, True):
                seq = data.TrainingSequence(root, dict(batch_size=B, step=1, device_jpeg=dj), {'image_size': S, 'bb_info_c_size': 6})
                f = BatchFeeder(seq, 1, 0, threads)
                for k in range(4):
                    f.load(k % 2)
                t0 = time.perf_counter()
                for k in range(10):
                    f.load(k % 2)
                dt = (time.perf_counter() - t0) / 10
                print('threads %2d  device_jpeg %-5s  %.1f ms per 40-image batch = %.0f img/s' % (threads, dj, dt * 1e3, B / dt), flush=True)
                f.close()
        # phases of the split, one thread
        names = sorted(os.listdir(root))[:40]
        names = [n for n in names if n.endswith('.jpg')]
        t0 = time.perf_counter(); datas = [open(os.path.join(root, n), 'rb').read() for n in names]; t1 = time.perf_counter()
        infos = [jpeg.parse(d) for d in datas]; t2 = time.perf_counter()
        plan = jpeg.BatchPlan(infos); t3 = time.perf_counter()
        buf = np.empty(plan.total_coefs, np.int16)
        for i in range(len(names)):
            jpeg.entropy_decode(datas[i], infos[i], buf[plan.coef_off[i]:plan.coef_off[i] + int(infos[i].total_coefs)])
        t4 = time.perf_counter()
        print('one thread, %d images: read %.1f ms, parse %.1f, plan %.1f, entropy decode %.1f (%.2f ms/img); file bytes %.1f MB, coefficients %.1f MB'
              % (len(names), (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t3) * 1e3 / len(names),
                 sum(len(d) for d in datas) / 1e6, plan.total_coefs * 2 / 1e6))
        t0 = time.perf_counter()
        for n in names:
            data._pil_loader(os.path.join(root, n))
        print('one thread, Pillow full decode: %.2f ms/img' % ((time.perf_counter() - t0) * 1e3 / len(names)))


if __name__ == '__main__':
    main()

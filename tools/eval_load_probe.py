"""Phases of one read-ahead load of FaceDetector._detect_files (16 JPEGs): read, parse, plan, pinned buffer, Huffman decode."""
import os, sys, time, tempfile

def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np, torch
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    from face_vijnana_yolov3_amd import jpeg
    torch.cuda.init()
    with tempfile.TemporaryDirectory() as root:
        rng = np.random.default_rng(0)
        sizes = [(768, 1024), (1024, 768), (720, 1280), (600, 800)]
        files = []
        for k in range(64):
            h, w = sizes[k % 4]
            lo = rng.integers(0, 256, (h // 16 + 1, w // 16 + 1, 3), dtype=np.uint8)
            files.append(os.path.join(root, 'img_%04d.jpg' % k))
            Image.fromarray(lo).resize((w, h), Image.BICUBIC).save(files[-1], quality=90)
        pool = ThreadPoolExecutor(max_workers=16)
        for rep in range(8):
            chunk = files[(rep % 4) * 16:(rep % 4) * 16 + 16]
            t = [time.perf_counter()]
            datas = list(pool.map(lambda f: open(f, 'rb').read(), chunk)); t.append(time.perf_counter())
            infos = [jpeg.parse(d) for d in datas]; t.append(time.perf_counter())
            plan = jpeg.BatchPlan(infos); t.append(time.perf_counter())
            buf = torch.empty(plan.total_coefs, dtype=torch.int16); t.append(time.perf_counter())
            buf = buf.pin_memory(); t.append(time.perf_counter())
            view = buf.numpy()
            list(pool.map(lambda i: jpeg.entropy_decode(datas[i], infos[i], view[plan.coef_off[i]:plan.coef_off[i] + int(infos[i].total_coefs)]), range(16)))
            t.append(time.perf_counter())
            print('rep %d: read %.2f parse %.2f plan %.2f empty %.2f pin %.2f decode %.2f ms  (%.1f MB)' %
                  ((rep,) + tuple((t[i + 1] - t[i]) * 1e3 for i in range(6)) + (plan.total_coefs * 2 / 1e6,)), flush=True)
            del buf, view


if __name__ == '__main__':
    main()

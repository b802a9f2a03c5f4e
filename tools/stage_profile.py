"""Device half of the input path alone (DeviceStager.stage on a 40-image batch, 10 times) -- run under rocprofv3 --kernel-trace
--stats to see what fv_jpeg_reconstruct_batch / fv_letterbox_batch cost per batch; also prints the wall time per staged batch."""
import os, sys, time, tempfile

def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np, torch
    from PIL import Image
    from face_vijnana_yolov3_amd import data
    from face_vijnana_yolov3_amd.engine import Engine
    from face_vijnana_yolov3_amd.face_detection import BatchFeeder, DeviceStager
    B, S = 40, 416
    eng = Engine(0)
    with tempfile.TemporaryDirectory() as root:
        rng = np.random.default_rng(0)
        sizes = [(768, 1024), (1024, 768), (720, 1280), (600, 800)]
        rows = []
        for k in range(B):
            h, w = sizes[k % 4]
            lo = rng.integers(0, 256, (h // 16 + 1, w // 16 + 1, 3), dtype=np.uint8)
            Image.fromarray(lo).resize((w, h), Image.BICUBIC).save(os.path.join(root, 'img_%04d.jpg' % k), quality=90)
            rows.append([k, 'img_%04d.jpg' % k, 1, 10.0, 10.0, 50.0, 60.0])
        import pandas as pd
        pd.DataFrame(rows, columns=data.CSV_COLUMNS).to_csv(os.path.join(root, 'training.csv'), index=False)
        for dj in (True, False):
            seq = data.TrainingSequence(root, dict(batch_size=B, step=1, device_jpeg=dj), {'image_size': S, 'bb_info_c_size': 6})
            f = BatchFeeder(seq, 1, 0, 16)
            st = DeviceStager(eng, S)
            items = [f.load(0) for _ in range(3)]
            for it in items:
                st.use(st.stage(it))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(9):
                st.use(st.stage(items[k % 3]))
            torch.cuda.synchronize()
            nbytes = items[0][0][1].numel() * 2 if dj else items[0][0][0].numel()
            print('device_jpeg %-5s: %.3f ms per staged 40-image batch (H2D of %.1f MB + kernels)' % (dj, (time.perf_counter() - t0) / 9 * 1e3, nbytes / 1e6), flush=True)
            f.close()


if __name__ == '__main__':
    main()

"""FaceDetector.test() over 512 synthetic JPEGs, three passes at the default eval batch -- to be run under
`rocprofv3 --kernel-trace --memory-copy-trace`: the timeline of the last pass (tools/test_loop_trace_summary.py) shows how busy the
device is and what it waits for."""
import os, sys, time, tempfile, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    from PIL import Image
    import bench
    from face_vijnana_yolov3_amd import face_detection
    n_img = 512
    rng = np.random.default_rng(0)
    sizes = [(768, 1024), (1024, 768), (720, 1280), (600, 800)]
    base = []
    for k in range(16):
        h, w = sizes[k % len(sizes)]
        lo = rng.integers(0, 256, (h // 16 + 1, w // 16 + 1, 3), dtype=np.uint8)
        base.append(Image.fromarray(lo).resize((w, h), Image.BICUBIC))
    with tempfile.TemporaryDirectory() as root:
        for k in range(n_img):
            base[k % 16].save(os.path.join(root, 'img_%04d.jpg' % k), quality=90)
        conf = {'mode': 'test', 'raw_data_path': root, 'test_path': root, 'output_file_path': os.path.join(root, 'solution_fd.csv'),
                'multi_gpu': False, 'num_gpus': 1, 'yolov3_base_model_load': False, 'model_loading': False,
                'hps': dict(bench.HPS, epochs=1, step=1, batch_size=40, face_conf_th=0.5, nms_iou_th=0.5, num_cands=60),
                'nn_arch': {'image_size': 416, 'bb_info_c_size': 6, 'head': 'single'}}
        face_detection.DEBUG = False
        with contextlib.redirect_stdout(io.StringIO()):
            fd = face_detection.FaceDetector(conf, 0)
        d = fd.model.layers[-1]
        fd.model.params[d['w_off']:d['beta_off']] *= 0.05
        fd.model.params[d['beta_off']] = 0.3; fd.model.params[d['beta_off'] + 5] = 0.3
        for p in range(3):
            t0 = time.perf_counter(); fd.test(); dt = time.perf_counter() - t0
            print('pass %d: %.1f img/s (%.1f ms)' % (p, n_img / dt, dt * 1e3), flush=True)


if __name__ == '__main__':
    main()

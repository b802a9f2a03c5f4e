"""cProfile of FaceDetector.test() at eval_batch_size 16 on synthetic JPEGs: where the host time of the loop goes."""
import cProfile, io, os, pstats, sys, tempfile, time, contextlib

def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np
    from PIL import Image
    from face_vijnana_yolov3_amd import face_detection
    n_img, S = int(os.environ.get('N_IMG', 128)), 416
    with tempfile.TemporaryDirectory() as root:
        rng = np.random.default_rng(0)
        sizes = [(768, 1024), (1024, 768), (720, 1280), (600, 800)]
        for k in range(n_img):
            h, w = sizes[k % 4]
            lo = rng.integers(0, 256, (h // 16 + 1, w // 16 + 1, 3), dtype=np.uint8)
            Image.fromarray(lo).resize((w, h), Image.BICUBIC).save(os.path.join(root, 'img_%04d.jpg' % k), quality=90)
        conf = {'mode': 'test', 'raw_data_path': root, 'test_path': root, 'output_file_path': os.path.join(root, 'solution_fd.csv'),
                'multi_gpu': False, 'num_gpus': 1, 'yolov3_base_model_load': False, 'model_loading': False,
                'hps': dict(lr=1e-4, beta_1=0.99, beta_2=0.99, decay=0.0, epochs=1, step=1, batch_size=40, face_conf_th=0.5, nms_iou_th=0.5,
                            num_cands=60, loader_threads=16, eval_batch_size=int(os.environ.get('EVAL_BS', 16))),
                'nn_arch': {'image_size': S, 'bb_info_c_size': 6}}
        face_detection.DEBUG = False
        with contextlib.redirect_stdout(io.StringIO()):
            fd = face_detection.FaceDetector(conf, 0)
        d = fd.model.layers[-1]
        fd.model.params[d['w_off']:d['beta_off']] *= 0.05
        fd.model.params[d['beta_off']] = 0.3; fd.model.params[d['beta_off'] + 5] = 0.3
        fd.test()
        t0 = time.perf_counter(); fd.test(); dt = time.perf_counter() - t0
        print('test(): %d images, %.1f img/s (%.1f ms per batch)' % (n_img, n_img / dt, dt / (n_img / conf['hps']['eval_batch_size']) * 1e3), flush=True)
        pr = cProfile.Profile(); pr.enable(); fd.test(); pr.disable()
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28); print(s.getvalue()[:6000])


if __name__ == '__main__':
    main()

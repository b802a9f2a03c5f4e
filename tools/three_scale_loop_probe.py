"""Where FaceDetector.test() with the three-scale head spends a batch, at eval batch 16 and 32 (709 -> 537 img/s)."""
import os, sys, time


def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np
    import torch
    import bench
    from face_vijnana_yolov3_amd import face_detection
    face_detection.DEBUG = False
    conf = {'mode': 'test', 'raw_data_path': '.', 'test_path': '.', 'output_file_path': '/tmp/x.csv', 'multi_gpu': False, 'num_gpus': 1,
            'yolov3_base_model_load': False, 'model_loading': False,
            'hps': dict(bench.HPS, epochs=1, step=1, batch_size=40, face_conf_th=0.5, nms_iou_th=0.5, num_cands=60),
            'nn_arch': {'image_size': 416, 'bb_info_c_size': 6, 'head': 'three_scale'}}
    fd = face_detection.FaceDetector(conf, 0)
    for d in fd.model.layers:
        if not d['has_bn']:
            fd.model.params[d['w_off']:d['beta_off']] *= 0.05
            fd.model.params[d['beta_off'] + 4:d['beta_off'] + d['cout']:6] = -2.0

    def ev_ms(fn, reps=5):
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    from face_vijnana_yolov3_amd.yolov3 import decode_nms_batch
    for B in (8, 16, 32, 64):
        x = torch.rand((B, 416, 416, 3), device='cuda')
        fwd = ev_ms(lambda: fd.model.predict_device(x))
        ys = fd.model.predict_device(x)
        dec = ev_ms(lambda: decode_nms_batch(fd.model.ctx, ys[0], ys[1], ys[2], (416, 416), (416, 416), obj_thresh=0.5, nms_thresh=0.5))
        res = decode_nms_batch(fd.model.ctx, ys[0], ys[1], ys[2], (416, 416), (416, 416), obj_thresh=0.5, nms_thresh=0.5)
        cnt = res['count'].cpu().numpy()
        t0 = time.perf_counter()
        for _ in range(5):
            launched = fd._detect_launch(x)
            out = fd._detect_collect(launched)
        whole = (time.perf_counter() - t0) / 5 * 1e3
        launched = fd._detect_launch(x); torch.cuda.synchronize()
        t0 = time.perf_counter(); fd._detect_collect(launched); host = (time.perf_counter() - t0) * 1e3
        print('B=%2d  forward %.2f ms (%.3f /img)  decode+nms %.2f ms (%.3f /img)  launch+collect %.2f ms (%.3f /img)  collect alone (host) %.2f ms  candidates/img %d..%d'
              % (B, fwd, fwd / B, dec, dec / B, whole, whole / B, host, cnt.min(), cnt.max()), flush=True)


if __name__ == '__main__':
    main()

import os, sys, torch

def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from face_vijnana_yolov3_amd.engine import Engine
    eng = Engine(0); eng.init_synthetic(seed=7)
    x = torch.rand((1,416,416,3), device='cuda')
    for _ in range(5): eng.predict_device(x)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): eng.predict_device(x)
    e1.record(); torch.cuda.synchronize()
    print('bs1 predict ms', e0.elapsed_time(e1)/50)
    eng.ctx.profile(True)
    for _ in range(10): eng.predict_device(x)
    p = eng.ctx.profile_collect(); eng.ctx.profile(False)
    tot=0
    for k,v in sorted(p.items(), key=lambda kv:-kv[1]['ms']):
        print('%-34s launches/img %5.1f  ms/img %.4f  us/launch %.2f' % (k, v['launches']/10, v['ms']/10, v['ms']/v['launches']*1e3)); tot+=v['ms']/10
    print('sum of kernel ms/img', tot)
    if os.environ.get('BS1_MAPS'):       # library map of this process: attributes the frames of an exit-time fault
        with open(os.environ['BS1_MAPS'], 'w') as f:
            for ln in open('/proc/self/maps'):
                if 'r-xp' in ln or 'r--p' in ln:
                    f.write(ln)
    if os.environ.get('BS1_CLOSE'):      # release the context explicitly, before interpreter shutdown
        eng.ctx.close()


if __name__ == '__main__':
    main()

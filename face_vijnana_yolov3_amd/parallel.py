"""Data-parallel training: one process per GPU, gradients all-reduced over RCCL/xGMI.

Replaces keras.utils.multi_gpu_model (reference face_detection.py:369, 612-619): every rank runs
the full model on its 40-image slice with PER-RANK BatchNorm statistics (= the reference's
per-tower statistics), the loss is the mean over the merged batch, so the gradient is the mean of
the per-rank gradients.  Gradients live in one flat fp32 vector; as the backward pass finishes a
layer the C ABI reports its range (fv_bucket_fn) and ranges are coalesced into buckets of
>= bucket_bytes that are all-reduced while backward continues.  Where the collectives run is
`comm_mode` (FV_COMM_STREAM): the default 'auto' starts in 'wg' -- a blocking all_reduce enqueued on
the library's weight-gradient stream, right behind the kernels that made the bucket, beside the
data-gradient chain on the compute stream -- and keeps it unless a measurement over the first steps
of the job shows another mode more than 1 % faster (DataParallelTrainer below).  The compute stream
joins the communication once per step, before Adam.  xGMI is
point-to-point, so few large collectives beat many small ones: default bucket 32 MiB
(162.56 MB of gradients -> 5-6 collectives per step).  Adam then runs redundantly on every rank
(bit-identical weights, no broadcast).

Uneven slices (multi_gpu_model gives the remainder of a short last batch to the last tower): the
merged-batch MSE is sum_r n_r/N * loss_r, so every rank hands its share n_r/N to the step as
`loss_weight` (fv_train_step scales dL/dy with it in the loss kernel: the gradients arrive pre-scaled,
no pass over the 162 MB vector) and the ranks SUM; a batch with fewer images than ranks is skipped on
ALL ranks (`slice_batch` returns None everywhere), never on some -- a rank that stayed out of a
collective would hang the others."""
import os
import shutil

import torch
import torch.distributed as dist


class BucketReducer(object):
    """Coalesce completed gradient ranges (arriving in descending offset order) into contiguous
    buckets and all-reduce(mean) each.  Backend-agnostic: `launch(view)` performs the collective
    (tests drive it with gloo on CPU tensors)."""

    def __init__(self, flat, world_size, bucket_bytes, launch):
        self.flat = flat
        self.world = world_size
        self.bucket_elems = max(1, bucket_bytes // flat.element_size())
        self.launch = launch
        self.reset()

    def reset(self):
        self.lo = self.hi = None
        self.launched = []

    def on_range(self, off, cnt):
        if self.hi is None:
            self.lo, self.hi = off, off + cnt
        else:
            if off + cnt != self.lo:
                raise RuntimeError('gradient ranges must arrive contiguously in descending order '
                                   '(got [%d,%d) after lo=%d)' % (off, off + cnt, self.lo))
            self.lo = off
        if self.hi - self.lo >= self.bucket_elems:
            self.flush()

    def flush(self):
        if self.hi is None:
            return
        view = self.flat[self.lo:self.hi]
        self.launch(view)
        self.launched.append((self.lo, self.hi))
        self.lo = self.hi = None


class DataParallelTrainer(object):
    def __init__(self, engine, world_size=None, rank=None, bucket_bytes=32 << 20, force_bucket_path=False, comm_mode=None):
        self.eng = engine
        self.world = int(world_size if world_size is not None else os.environ.get('WORLD_SIZE', 1))
        self.rank = int(rank if rank is not None else os.environ.get('RANK', 0))
        self.bucket_bytes = bucket_bytes
        self.comm = None
        self.reducer = None
        self.bucketed = self.world > 1 or force_bucket_path  # force: exercise the stream/bucket path on 1 GPU
        # Where the collectives run (FV_COMM_STREAM):
        #   'wg'   on the library's weight-gradient (side) stream, where the gradients are made: fv_bucket_fn fires as soon as a
        #          range's weight-gradient kernels are in that stream's queue (fv_set_bucket_on_side) and a BLOCKING all_reduce is
        #          enqueued there -- it blocks the side stream only, the data-gradient chain on the compute stream runs on beside
        #          it and no event passes between the compute stream and the communication until the join at the end of the
        #          backward pass.  Costs 0-0.4 ms per step at one rank (52.2 against 52.2, 52.1 against 52.0, 53.0 against 52.6 for
        #          'main' on three boxes; tools/dp_trace.py, bench.py);
        #   'pg'   dist.all_reduce(async_op=True): the collective runs on the process group's OWN stream, ordered after the
        #          compute stream's work so far by the backend, and the compute stream waits for all of them before Adam -- the
        #          overlap with the rest of the backward pass without a second hop;
        #   'side' a communication stream of this trainer (compute -> event -> comm stream -> blocking all_reduce -> backend stream
        #          and back): the round-1/2 form; FV_COMM_PRIORITY sets its priority (-1 = high);
        #   'main' blocking all_reduce on the compute stream (no overlap).
        # Measured on one MI355X with a world-size-1 nccl group (no RCCL kernel runs there: the stream / event traffic alone;
        # tools/dp_overhead2.py, tools/dp_trace.py; plain step 52.2 ms): 'wg' 52.2, 'main' 52.2, 'pg' 55.6, 'side' the same as
        # 'pg' (worse at high priority or with GPU_MAX_HW_QUEUES=8), independent of the number of buckets.
        #   'auto' (default) starts in 'wg'.  With more than one rank the first steps of the job time 'wg', 'main' and 'pg' over
        #          AUTO_STEPS steps each (max over ranks, so every rank decides alike) and 'wg' is kept unless another mode is
        #          more than 1 % faster -- self.auto_report says what was measured.  At one rank there is nothing to overlap
        #          (RCCL launches no kernel for a one-rank all-reduce) and no contest: 'wg'.
        # 'pg' / 'side' put 60 us bubbles in front of two dozen kernels of the backward pass and stretch the rest (rocprofv3
        # timeline of tools/dp_trace.py: 55.6 against 52.2 ms per step at one rank); whether real communication time hidden on a
        # third stream is worth that on a node is what the calibration decides.
        self.comm_mode = comm_mode or os.environ.get('FV_COMM_STREAM', 'auto')
        if self.comm_mode not in ('pg', 'side', 'main', 'auto', 'wg'):
            raise ValueError("FV_COMM_STREAM must be 'auto', 'wg', 'pg', 'side' or 'main'")
        self.auto_report = None
        self._auto_k = None
        self._auto_ms = []
        self._auto_skip = False
        self.auto_abandoned = {}
        if self.comm_mode == 'auto':
            self.comm_mode = 'wg'
            # resolved in train_on_batch (needs a process group to be worth measuring); a trainer without the bucket path has
            # no collectives and nothing to decide
            self._auto_k = 0 if self.bucketed else None
        self._works = []
        self.collectives_launched = 0
        if self.bucketed:
            if self.comm_mode == 'side':
                self.comm = torch.cuda.Stream(device=engine.dev, priority=int(os.environ.get('FV_COMM_PRIORITY', '0')))
            if not engine.ctx.side_stream():
                raise RuntimeError('the context has no side stream: comm_mode wg needs fv_side_stream(ctx)')
            self.wg_stream = torch.cuda.ExternalStream(engine.ctx.side_stream(), device=engine.dev)
            engine.ensure_optimizer()
            self.reducer = BucketReducer(engine.grads, self.world, bucket_bytes, self._launch)
        self._own_group = False
        if self.world > 1:
            if not dist.is_initialized():
                self._own_group = True
                os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
                os.environ.setdefault('MASTER_PORT', '29500')
                backend = os.environ.get('FV_DIST_BACKEND', 'nccl')  # 'nccl' IS RCCL on ROCm; gloo only to rehearse on one GPU
                if backend == 'nccl':
                    dist.init_process_group('nccl', rank=self.rank, world_size=self.world, device_id=engine.dev)
                else:
                    dist.init_process_group(backend, rank=self.rank, world_size=self.world)
        # The collectives run whenever a process group exists -- also a group of ONE rank (a `nccl` group of world size 1 on a
        # single GPU sends every bucket through RCCL on the comm stream: the hardware rehearsal of the N > 1 path).
        self.collective = dist.is_available() and dist.is_initialized()
        if self.collective and dist.get_world_size() != self.world:
            raise RuntimeError('process group has %d ranks, trainer was told %d' % (dist.get_world_size(), self.world))
        if self.collective:
            # identical start on every rank
            dist.broadcast(engine.params, 0)
            dist.broadcast(engine.state, 0)

    def _launch(self, view):
        """One bucket: all-reduce(SUM).  The gradients arrive scaled by n_rank / n_total (loss_weight of the step), so the
        sum over the ranks is the gradient of the merged-batch mean."""
        if self.comm_mode == 'wg':
            # the range was reported from the side stream's queue (fv_set_bucket_on_side): its weight-gradient kernels are ahead
            # of anything enqueued there now.  Without the overlap the library reports ranges on the compute stream itself.
            if not self.eng.ctx.overlap:
                return self._collective(view)
            with torch.cuda.stream(self.wg_stream):
                self._collective(view)      # blocking for the SIDE stream only
            return
        if self.comm_mode == 'side':
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.eng.dev))
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(ev)
                self._collective(view)
            return
        self._collective(view, async_op=self.comm_mode == 'pg')

    _probe = None      # test hook: called with the bucket view at the point of the collective, on the stream the collective uses

    def _collective(self, view, async_op=False):
        if self._probe is not None:
            self._probe(view)
        if not self.collective:
            return
        if async_op:
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))
        else:
            dist.all_reduce(view, op=dist.ReduceOp.SUM)
        self.collectives_launched += 1

    AUTO_WARM, AUTO_STEPS = 3, 8
    AUTO_MODES = ('wg', 'main', 'pg')
    AUTO_MARGIN = 0.01       # 'wg' is kept unless another mode is more than this fraction faster
    AUTO_ABANDON = 3.0       # a mode whose first step takes more than this many 'wg' steps is dropped without being timed

    def _auto_tick(self):
        """comm_mode 'auto': called at the top of every step until decided.  AUTO_WARM steps warm up in 'wg'; then every mode
        of AUTO_MODES in turn runs one untimed step and AUTO_STEPS timed ones (host clock around a device synchronisation);
        the times are max-reduced over the ranks and 'wg' is kept unless another mode beats it by more than AUTO_MARGIN.
        Every step of the calibration is an ordinary training step.  One rank: no contest, 'wg'."""
        import time
        if not self.collective:                  # nothing to overlap without a group
            self.comm_mode, self._auto_k = 'main', None
            return
        if self.world == 1:
            self.comm_mode, self._auto_k = 'wg', None
            self.auto_report = dict(chosen='wg', skipped='one rank: RCCL runs no kernel for a one-rank all-reduce, nothing to compare')
            return
        W, S, k = self.AUTO_WARM, self.AUTO_STEPS, self._auto_k
        per = S + 1                               # steps per mode: one warm-up + S timed
        if k >= W:
            j, r = divmod(k - W, per)
            if r == 0 and j > 0:                  # mode j-1 has run its S timed steps (or was dropped after its first)
                torch.cuda.synchronize(self.eng.dev)
                if self._auto_skip:
                    self._auto_skip = False
                else:
                    self._auto_ms.append((time.perf_counter() - self._auto_t0) / S * 1e3)
            if r == 0 and j == len(self.AUTO_MODES):
                t = torch.tensor(self._auto_ms, dtype=torch.float64, device=self.eng.dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                ms = dict(zip(self.AUTO_MODES, (float(v) for v in t.cpu())))
                best = min(ms, key=ms.get)
                self.comm_mode = best if ms[best] < ms['wg'] * (1.0 - self.AUTO_MARGIN) else 'wg'
                self.auto_report = dict(ms_per_step={m: round(v, 3) for m, v in ms.items()}, chosen=self.comm_mode, steps_each=S,
                                        rule="'wg' unless another mode is more than %g %% faster" % (100 * self.AUTO_MARGIN))
                if self.auto_abandoned:
                    self.auto_report['dropped_after_first_step_ms'] = dict(self.auto_abandoned)
                self._auto_k = None
                return
            if r == 0:
                self.comm_mode = self.AUTO_MODES[j]
                torch.cuda.synchronize(self.eng.dev)
                self._auto_tw = time.perf_counter()
            elif r == 1:
                torch.cuda.synchronize(self.eng.dev)
                self._auto_t0 = time.perf_counter()
                if j > 0:
                    # A mode whose FIRST step is several times slower than a 'wg' step is not a candidate: drop it after that one step
                    # instead of timing eight more (two ranks rehearsing on one GPU with GPU_MAX_HW_QUEUES=8: 'pg' took 17 s per
                    # step -- more streams than hardware queues left for two processes).  Decided on max-reduced numbers: alike everywhere.
                    t = torch.tensor([(self._auto_t0 - self._auto_tw) * 1e3, self._auto_ms[0]], dtype=torch.float64, device=self.eng.dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    warm_ms, wg_ms = (float(v) for v in t.cpu())
                    if warm_ms > self.AUTO_ABANDON * wg_ms:
                        self._auto_ms.append(warm_ms)
                        self.auto_abandoned[self.AUTO_MODES[j]] = round(warm_ms, 3)
                        k = W + (j + 1) * per - 1              # the next tick is the next mode's r == 0 (or the decision)
                        self.comm_mode = 'wg'                  # this step itself runs in the trusted mode
                        self._auto_skip = True
        self._auto_k = k + 1

    @property
    def calibrating(self):
        """True while comm_mode 'auto' is still measuring (bench.py keeps those steps out of its timed region)."""
        return self._auto_k is not None

    def allreduce_ms(self):
        """Cost of one step's gradient + BN-state collectives when NOTHING overlaps them: the buckets of the last step are
        all-reduced again back to back on the compute stream between two events (leaves the gradients scaled; call it after
        the measurements that need them).  0 without a process group."""
        if not (self.collective and self.bucketed and self.reducer.launched):
            return 0.0
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(self.eng.dev)
        e0.record()
        for lo, hi in self.reducer.launched:
            dist.all_reduce(self.eng.grads[lo:hi], op=dist.ReduceOp.SUM)
        dist.all_reduce(self.eng.state, op=dist.ReduceOp.SUM)
        self.eng.ctx.scale(self.eng.state, 1.0 / self.world)
        e1.record(); torch.cuda.synchronize(self.eng.dev)
        return e0.elapsed_time(e1)

    def train_on_batch(self, x, y, lr, beta_1, beta_2, decay=0.0, weight=None):
        """weight: this rank's share n_rank / n_total of the merged batch (default 1 / world)."""
        eng = self.eng
        if not self.bucketed:
            return eng.train_on_batch(x, y, lr, beta_1, beta_2, decay)
        if self._auto_k is not None:
            self._auto_tick()
        self._weight = float(weight) if weight is not None else 1.0 / self.world
        on_side = self.comm_mode == 'wg' and eng.ctx.overlap
        eng.ctx.set_bucket_on_side(on_side)       # per step: several trainers may share one context
        self.reducer.reset()
        try:
            loss = eng.forward_backward(x, y, on_bucket=self.reducer.on_range, loss_weight=self._weight)
        finally:
            eng.ctx.set_bucket_on_side(False)     # a later direct forward_backward(on_bucket=...) gets stream-ordered callbacks again
        self.reducer.flush()
        main = torch.cuda.current_stream(eng.dev)
        # BN moving statistics: the reference's towers race on shared variables (undefined order); we keep ranks identical
        # by averaging (parity unpinned): SUM over the ranks, then fv_scale by 1 / world on the compute stream
        if self.comm_mode == 'side':
            with torch.cuda.stream(self.comm):
                ev = torch.cuda.Event(); ev.record(main)
                self.comm.wait_event(ev)
                if self.collective:
                    dist.all_reduce(eng.state, op=dist.ReduceOp.SUM)
                    self.collectives_launched += 1
            main.wait_stream(self.comm)
        else:
            if on_side:       # the last bucket went out after the step had joined the side stream: join it again
                main.wait_stream(self.wg_stream)
            if self.collective:
                if self.comm_mode == 'pg':
                    self._works.append(dist.all_reduce(eng.state, op=dist.ReduceOp.SUM, async_op=True))
                else:
                    dist.all_reduce(eng.state, op=dist.ReduceOp.SUM)
                self.collectives_launched += 1
            for w in self._works:           # the compute stream waits for every collective of this step (no host wait with nccl)
                w.wait()
            self._works = []
        if self.world > 1:
            eng.ctx.scale(eng.state, 1.0 / self.world)
        eng.adam_step(lr, beta_1, beta_2, decay)
        return loss

    def merged_loss(self, loss, weight=None):
        """Loss of the merged batch, sum_r n_r/N * loss_r -- what Keras prints for a multi_gpu_model step (one loss over the
        concatenated tower outputs, fd.py:366-371).  Host sync; a collective when a group exists (every rank must call it)."""
        if not self.collective or self.world == 1:
            return float(loss.item())
        t = loss.detach().double().reshape(1) * (float(weight) if weight is not None else 1.0 / self.world)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def barrier(self):
        if self.collective:
            dist.barrier()

    def max_over_ranks(self, value):
        if not self.collective:
            return value
        t = torch.tensor([value], dtype=torch.float64, device=self.eng.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def shutdown(self):
        if self.collective and dist.is_initialized():
            dist.barrier()
            if self._own_group:
                dist.destroy_process_group()


def launch_ranks(n, target, extra_env=None, on_stdout_line=None):
    """Start n ranks (one process per GPU, rendezvous on 127.0.0.1 over a free port) as FRESH children through
    torch.distributed.run and return the launcher's exit code.  `target` is what follows the launcher's own options:
    ['script.py', args...] or ['-m', 'package.module', args...].  Must be called before this process makes any GPU call (the
    children own the devices).  With on_stdout_line the children's stdout is piped and every line handed to it; otherwise they
    inherit this process's descriptors."""
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(int(n)), '--master-addr', '127.0.0.1',
           '--master-port', str(port)] + list(target)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # this pool's driver only supports dmabuf IPC (RCCL needs it)
    env.update(extra_env or {})
    print('starting %d ranks: %s' % (n, ' '.join(cmd)), file=sys.stderr, flush=True)
    if on_stdout_line is None:
        return subprocess.call(cmd, env=env)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1)
    for line in proc.stdout:
        on_stdout_line(line)
    return proc.wait()


def slice_batch(n, world_size, rank):
    """Contiguous tower slice [lo, hi) of an n-image batch, as keras.utils.multi_gpu_model cuts it
    (fd.py:369: n // world each, the remainder to the last tower), with this rank's weight n_r / n in the
    merged-batch mean.  None on EVERY rank when n < world (some tower would be empty)."""
    per = n // world_size
    if per == 0:
        return None
    lo = rank * per
    hi = n if rank == world_size - 1 else lo + per
    return lo, hi, (hi - lo) / float(n)


def ensure_process_group(device=None):
    """evaluate()/test() under torchrun need a barrier but no trainer: join the default group."""
    if int(os.environ.get('WORLD_SIZE', 1)) > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        backend = os.environ.get('FV_DIST_BACKEND', 'nccl')
        if backend == 'nccl' and device is not None:
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)


def _barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def reset_dir_before_shards(path, rank):
    """Rank 0 empties `path`, every rank waits for that, then creates it: no rank can lose files it wrote
    to a directory another rank is still deleting."""
    if rank == 0:
        shutil.rmtree(path, ignore_errors=True)
        os.makedirs(path, exist_ok=True)
    _barrier()
    os.makedirs(path, exist_ok=True)


def part_path(path, world_size, rank):
    return path if world_size == 1 else '%s.rank%d' % (path, rank)


def merge_rank_files(path, world_size, rank, header_lines=0):
    """After every rank has written and closed part_path(path, ...): rank 0 concatenates the parts in rank
    order into `path` (file shards are contiguous, so this is the single-process row order), keeping the
    first part's `header_lines` only, and removes the parts.  Collective: all ranks must call it."""
    if world_size == 1:
        return
    _barrier()
    if rank == 0:
        with open(path, 'w') as out:
            for r in range(world_size):
                with open(part_path(path, world_size, r)) as f:
                    lines = f.readlines()
                out.writelines(lines if r == 0 else lines[header_lines:])
        for r in range(world_size):
            os.remove(part_path(path, world_size, r))
    _barrier()


def shard_files(file_names, world_size, rank):
    """evaluate/test are embarrassingly parallel over images: contiguous file-list shard per rank,
    no collective."""
    n = len(file_names)
    per = (n + world_size - 1) // world_size
    return file_names[rank * per:min(n, (rank + 1) * per)]

"""MI355X-native FaceDetector hot path (gfx950 HIP kernels behind a C ABI).

The product path never falls back to a CPU implementation: if libfv_hotpath.so is missing
or no MI355X is visible, calls raise."""
import os as _os

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  fv_train_step overlaps its
# weight-gradients with the data-gradient chain on a low-priority side stream of its context; a context created AFTER an RCCL
# communicator has existed in the process (DataParallelTrainer, or bench.py's world-size-1 rehearsal) found its side stream sharing a
# hardware queue with the compute stream: the three-scale step ran 31.5 -> 43 ms and the 608 x 608 step 45.2 -> 51.7 ms, while contexts
# created before the communicator were unaffected (round 5, gpurun_out/r5_hist3.txt: 8 queues, a default-priority side stream or no
# overlap each restore the fresh-process time; this is what rounds 3 and 4 chased as the "late in the process" slowdown).  The variable
# is read when the HIP runtime initialises, so it only takes effect if this package is imported before the first GPU call; a value
# set by the user wins.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

__version__ = '0.1.0'

"""Numeric contract of the slim.h C API, as the Python side needs it.

Values follow /root/reference/include/slim.h:56 (SLIM_NOPTIONS), :177-182
(return codes), :185-209 (model / similarity / algorithm enums), :215-230
(option slots), :233-239 (debug bits).  Slots >= 11 of the option arrays are
unused by the reference; this engine assigns some of them (include/slim_gpu.h).
"""
import enum

SLIM_VERSION = "2.0"
SLIM_NOPTIONS = 40


class Status(enum.IntEnum):
    OK = 1
    ERROR_INPUT = -2
    ERROR_MEMORY = -3
    ERROR = -4


SLIM_OK = int(Status.OK)
SLIM_ERROR_INPUT = int(Status.ERROR_INPUT)
SLIM_ERROR_MEMORY = int(Status.ERROR_MEMORY)
SLIM_ERROR = int(Status.ERROR)


class Opt(enum.IntEnum):
    """Index into ioptions[] / doptions[]; -1 in a slot selects the default."""
    DBGLVL = 0
    NNBRS = 1
    SIMTYPE = 2
    NTHREADS = 3
    MAXNITERS = 4
    ALGO = 5
    ORDERED = 6
    L1R = 7
    L2R = 8
    OPTTOL = 9
    NRCMDS = 10
    # ---- engine extensions (include/slim_gpu.h); ignored by the reference ----
    GPU_COLBEGIN = 11   # first item column this call solves (default 0)
    GPU_COLEND = 12     # one past the last item column (default ncols)
    GPU_SEED = 13       # seed of the per-sweep visiting permutation (default 1)
    GPU_DEVICE = 14     # HIP device ordinal (default: current device)
    GPU_KERNEL = 15     # kernel selection, see slim_gpu.h (default auto)
    GPU_CLUSTER = 16    # tile kernels: workgroups per tile (default auto)
    GPU_HEAVYTILES = 17    # tile kernels: most expensive tiles solved first by larger clusters
    GPU_HEAVYCLUSTER = 18  # ... of this many workgroups (default auto)
    GPU_NGPUS = 19         # SLIM_Learn & co: shard the item columns over this many GPUs (default 1)
    GPU_SHARDCOUNT = 20    # SLIMGPU_Learn*: solve one shard of the requested columns ...
    GPU_SHARDINDEX = 21    # ... granules INDEX, INDEX + COUNT, ... of the cost-ordered work list


for _o in Opt:
    globals()["SLIM_OPTION_" + _o.name] = int(_o)

SLIM_MTYPE = {"slim": 0, "fslim": 1, "oslim": 2, "ofslim": 3}
SLIM_SIMTYPE = {"cos": 0, "jac": 1, "dotp": 2}
SLIM_ALGO = {"admm": 0, "cd": 1}

SLIM_DBG_INFO = 1
SLIM_DBG_TIME = 2
SLIM_DBG_PROGRESS = 4
SLIM_DBG_PROGRESS2 = 16
SLIM_DBG_MEMORY = 2048

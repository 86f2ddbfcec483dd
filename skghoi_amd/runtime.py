"""Process-level HIP runtime settings of the interaction head -- an explicit entry point, not an import side effect.

    from skghoi_amd import runtime
    runtime.configure()          # before the first HIP call of the process (before torch touches the GPU)

The one setting: GPU_MAX_HW_QUEUES, the number of hardware queues the HIP runtime opens per stream-priority class
(default 4) and deals to streams round-robin.  On MI355X / ROCm 7.2 a process that gets to FOUR busy queues of the normal
class runs the training step's chains of small dependent kernels ~40 % slower from then on (batch-4 bf16 step 1.40 ->
1.98 ms, same kernels; profiles/r03_hw_queue_count_and_step_time.txt, profiles/r04_hw_queues_*.txt); three keeps every
measured sequence of eval / graph-replay / training legs fast and costs the eval paths nothing.  The runtime reads the
variable once, at its first call: `configure()` after that point changes nothing and says so.

Every launch form goes through the same call with the same default -- a single process, a rank under
torch.distributed.run, a rank started by `bench.py --gpus N` -- so ranks never differ in this setting.  An explicit
GPU_MAX_HW_QUEUES in the environment is respected; values below three are known to crash the runtime's hipGraph path
(two-branch captured graphs), so the head then keeps its small-batch eval on individually enqueued kernels
(`graphs_allowed()`), with a warning.
"""
import os
import warnings

DEFAULT_HW_QUEUES = 3
_MARK = "SKG_HW_QUEUES_SET_BY_SKGHOI"      # carries the value this module exported: a child process can tell it from the user's
_STATE = dict(configured=False, effective=None, source=None)


def _hip_started():
    try:
        import torch
        return torch.cuda.is_initialized()
    except Exception:                                   # noqa: BLE001
        return False


def configure(hw_queues=None):
    """Sets GPU_MAX_HW_QUEUES for this process unless the user exported it.  hw_queues: None -> SKG_HW_QUEUES from the
    environment, else DEFAULT_HW_QUEUES; 0 -> leave the runtime's default.  Returns the info() record."""
    if hw_queues is None and _STATE["configured"]:
        return info()                                   # (an application's explicit choice stands; libraries calling again change nothing)
    if hw_queues is None:
        env = os.environ.get("SKG_HW_QUEUES")
        hw_queues = int(env) if env not in (None, "") else DEFAULT_HW_QUEUES
    user = os.environ.get("GPU_MAX_HW_QUEUES")
    if _STATE["configured"] and _STATE["source"].startswith("skghoi_amd.runtime.configure"):
        user = None                                     # (our own earlier setting is not the user's)
    inherited = False
    if user is not None and os.environ.get(_MARK) == user:
        # set by THIS module in a parent process (bench.py --gpus N starts its ranks, the world-size-1 RCCL child): our own
        # choice travelling through the environment, not the user's -- the record must say so
        user, inherited = None, True
    if user is not None:
        _STATE.update(configured=True, effective=user, source="environment (GPU_MAX_HW_QUEUES exported by the user)")
    elif _hip_started():
        warnings.warn("skghoi_amd.runtime.configure() called after the HIP runtime started: GPU_MAX_HW_QUEUES keeps the "
                      "runtime's default for this process (call it before the first GPU use)")
        _STATE.update(configured=True, effective=None, source="too late: HIP runtime already initialised")
    elif hw_queues:
        os.environ["GPU_MAX_HW_QUEUES"] = str(int(hw_queues))
        os.environ[_MARK] = str(int(hw_queues))         # (child processes: this value is ours, see above)
        _STATE.update(configured=True, effective=str(int(hw_queues)),
                      source="skghoi_amd.runtime.configure" + (" (in the launching process)" if inherited else ""))
    else:
        if inherited:
            os.environ.pop("GPU_MAX_HW_QUEUES", None); os.environ.pop(_MARK, None)
        _STATE.update(configured=True, effective=None, source="runtime default (configure(hw_queues=0))")
    return info()


def info():
    """What this process runs with: {'GPU_MAX_HW_QUEUES': value or None (runtime default, 4), 'source': ...}."""
    eff = _STATE["effective"] if _STATE["configured"] else os.environ.get("GPU_MAX_HW_QUEUES")
    src = _STATE["source"] if _STATE["configured"] else \
        ("environment" if os.environ.get("GPU_MAX_HW_QUEUES") else "runtime default (configure() not called)")
    return {"GPU_MAX_HW_QUEUES": eff, "source": src}


def graphs_allowed():
    """False when the process runs with fewer than three hardware queues per class: the runtime's hipGraph path has crashed
    there (captured graphs with two branches); the small-batch eval then enqueues its kernels one by one."""
    v = os.environ.get("GPU_MAX_HW_QUEUES")
    try:
        return v is None or int(v) >= 3 or int(v) <= 0
    except ValueError:
        return True

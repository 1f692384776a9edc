"""Multi-GPU sharding of the classification path: one process per GPU (torchrun), sequences
sharded, tables replicated, NO collective on the data path (every sequence is independent,
stag/classify.py:155-158).  torch.distributed is used only to bring the text rows back to
rank 0 in input order, which is what keeps the output TSV identical to the single-process one.
"""
import os


def world_rank():
    """(world_size, rank) of the current torch.distributed job, (1, 0) outside one."""
    try:
        import torch.distributed as dist
    except ImportError:
        return 1, 0
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def init_from_env():
    """Under a launcher that sets RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun), join the process
    group before anything else runs, so that world_rank() is the truth in every rank: without it every rank
    would take itself for rank 0, classify the whole input and write the same output files.  gloo is enough
    -- the group only carries result rows to rank 0 (the data path has no collective).  Returns True when
    this call created the group (the caller then destroys it with shutdown())."""
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 or "RANK" not in os.environ:
        return False
    import torch.distributed as dist
    if dist.is_initialized():
        return False
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(os.environ.get("STAG_DIST_BACKEND", "gloo"))
    return True


def shutdown():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def local_device():
    return int(os.environ.get("STAG_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def devices_from_env():
    """GPUs one process drives (one host thread + one handle per device): STAG_DEVICES="0,1,2,3",
    or "all".  Default: the single device of local_device()."""
    spec = os.environ.get("STAG_DEVICES", "").strip()
    if not spec:
        return [local_device()]
    if spec.lower() == "all":
        from . import _lib
        return list(range(max(_lib.device_count(), 1)))
    return [int(tok) for tok in spec.split(",") if tok.strip() != ""]


class DevicePool:
    """One process, N GPUs: the tables are packed once and replicated (one Database per device), ONE aligner
    and ONE parser feed batches -- contiguous ranges of the input, in input order -- to N worker threads, each
    of which owns a device.  The calls into libstag_hip release the GIL, so the devices run side by side;
    results come back per batch and are put in input order by the caller.  No collective: the path is
    embarrassingly parallel (stag/classify.py:155-158 has no cross-sequence state).

    submit(fn, *args) runs fn(database_of_some_device, *args) on the next free device and returns a future;
    at most `depth` batches per device are in flight, so a fast parser cannot run ahead of the GPUs without bound."""

    def __init__(self, make_database, devices, depth=2, own=True):
        """own=False: the databases belong to the caller, close() leaves them open."""
        self._own = own
        import queue
        import threading
        from concurrent.futures import Future
        self._Future = Future
        self.dbs = [make_database(d) for d in devices]
        self._q = queue.Queue(maxsize=max(1, depth * len(self.dbs)))
        self._threads = [threading.Thread(target=self._work, args=(db,), daemon=True) for db in self.dbs]
        for t in self._threads:
            t.start()

    def _work(self, db):
        while True:
            item = self._q.get()
            if item is None:
                return
            fut, fn, args = item
            if not fut.set_running_or_notify_cancel():
                continue
            try:
                fut.set_result(fn(db, *args))
            except BaseException as e:      # delivered to the caller through the future
                fut.set_exception(e)

    @property
    def first(self):
        return self.dbs[0]

    def submit(self, fn, *args):
        fut = self._Future()
        self._q.put((fut, fn, args))
        return fut

    def close(self):
        for _ in self._threads:
            self._q.put(None)
        for t in self._threads:
            t.join()
        if self._own:
            for db in self.dbs:
                db.close()


def owner_of_batch(batch_index, world):
    """Batches of a stream are dealt round-robin: rank = batch index mod world."""
    return batch_index % world


def gather_in_order(tagged_rows, error=None, error_batch=0):
    """tagged_rows: list of (batch_index, rows) produced by this rank.  Returns, on rank 0, the rows
    of all ranks flattened in batch order; on other ranks an empty list.  If any rank passes
    an exception in ``error`` every rank raises (so that no rank waits for a dead peer) the one
    from the earliest batch (``error_batch``), which is where the reference's single loop stops."""
    world, rank = world_rank()
    if world == 1:
        if error is not None:
            raise error
        return [r for _, rows in sorted(tagged_rows, key=lambda t: t[0]) for r in rows]
    import torch.distributed as dist
    errors = [None] * world
    dist.all_gather_object(errors, None if error is None else (error_batch, error))
    failed = [e for e in errors if e is not None]
    if failed:
        raise min(failed, key=lambda t: t[0])[1]
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(tagged_rows, gathered, dst=0)
    if rank != 0:
        return []
    merged = [t for part in gathered for t in part]
    return [r for _, rows in sorted(merged, key=lambda t: t[0]) for r in rows]



"""Row sharding over one process per GPU, collectives by RCCL inside the HIP library (no PyTorch).

The primal path shards by rows: every rank calls ``primal_fit`` on its own contiguous block of
X, y, s and the library exchanges, on its own stream, exactly four things:

    1. {sum s, sum s*y, n}                  all-reduce   global weight normalisation, c = 1 / (n_total (D+1))
    2. the tile-packed Hermitian block A||b all-reduce   identical normal equations on every rank (139 MB at D = 4096)
    3. the eigen-decomposition              rank 0 runs the tridiagonal eigensolver and broadcasts (lam, C); every rank
                                            back-transforms its own column block of the eigenvectors; all-gather
    4. the per-gamma error vectors          all-reduce   identical argmin everywhere (+ two score scalars)

Per-row outputs (loo_residuals, loo_leverage, loo_std, residuals) stay sharded.

Joining the ranks: rank 0 draws a communicator id (``Context.comm_unique_id``) and the others need its 128 bytes.
``exchange_unique_id`` passes them through a file in a directory all ranks see (one node: ``/tmp``); the file name is
keyed on the launcher's process id and the rendezvous port so concurrent or consecutive launches never meet.
``init_from_env`` does the whole thing from the environment a launcher such as ``torch.distributed.run`` (or
``bench.py``'s own spawner) sets: RANK, WORLD_SIZE, LOCAL_RANK, MASTER_PORT.
"""

from __future__ import annotations

import os
import time
from pathlib import Path

__all__ = ["row_shard", "exchange_unique_id", "init_from_env"]


def row_shard(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced row block [lo, hi) of rank ``rank``; blocks tile [0, n) in rank order."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return (n * rank) // world, (n * (rank + 1)) // world


def _rendezvous_files(key: str | None) -> tuple[Path, Path | None]:
    """(primary, secondary).  Primary: keyed on the launcher's pid - all ranks of one launch are children of one launcher
    process - the rendezvous port, the run id and the elastic restart count (a restarted group never reads the previous
    attempt's file).  Secondary (only without an explicit key): keyed on the port alone, for a launcher that puts an
    intermediate process between itself and the ranks.  Both payloads carry a timestamp and are accepted only while fresh."""
    base = Path(os.environ.get("NLS_RENDEZVOUS_DIR", "/tmp"))
    if key is not None:
        return base / f"nls_rccl_id_{key}", None
    port, run = os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none")
    attempt = os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
    return base / f"nls_rccl_id_{os.getppid()}_{port}_{run}_{attempt}", base / f"nls_rccl_id_port{port}_{run}_{attempt}"


def _launch_nonce(explicit_key: bool = False) -> bytes:
    """What the ranks of ONE launch share and a previous launch does not: NLS_RENDEZVOUS_NONCE when the launcher sets it, else the
    rendezvous address and port, the elastic run id and restart count - and, for an automatic (launcher-derived) name only, the launcher's
    process id.  Rank 0 writes it into the payload; a reader joins only a payload that carries ITS OWN nonce - so an id file that a dead
    launch left under the same explicit key (its post-barrier unlink never ran) is not joined by the next launch's ranks while they wait for
    their rank 0 to replace it.
    An explicit key is the rendezvous of ranks that need NOT be children of one launcher (several nodes on a shared NLS_RENDEZVOUS_DIR,
    per-node daemons of mpirun / srun, hand-started ranks): their parent pids differ, so the pid is not part of their nonce - only values
    every rank of the launch has (two launches that share key, address, port and run id must set NLS_RENDEZVOUS_NONCE)."""
    explicit = os.environ.get("NLS_RENDEZVOUS_NONCE")
    if explicit:
        return explicit.encode()
    port, run = os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none")
    attempt = os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
    if explicit_key:
        return f"{os.environ.get('MASTER_ADDR', 'local')}_{port}_{run}_{attempt}".encode()
    return f"{os.getppid()}_{port}_{run}_{attempt}".encode()


def _fresh_seconds() -> float:
    """How long a published id stays joinable (NLS_RENDEZVOUS_FRESH_SECONDS, read at call time; default 900 s)."""
    return float(os.environ.get("NLS_RENDEZVOUS_FRESH_SECONDS", "900"))


def _publish(path: Path, payload: bytes) -> None:
    """Atomic publish: a private temporary file (O_EXCL | O_NOFOLLOW, mode 0600: never through a planted symlink), then rename."""
    tmp = path.with_name(f"{path.name}.tmp{os.getpid()}")  # (not with_suffix: a key such as "job.1" keeps its last dotted part)
    try:
        os.unlink(tmp)
    except FileNotFoundError:
        pass
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
    try:
        os.write(fd, payload)
    finally:
        os.close(fd)
    os.replace(tmp, path)


def _read_fresh(path: Path, explicit_key: bool) -> bytes | None:
    """The 128-byte id of a payload `id || repr(time) || "|" || nonce`.  Either kind of file could be a leftover of a launch that died
    between publishing and the post-barrier unlink.  A file under an explicit key is the caller's own rendezvous, joined whatever its
    age (a rank may arrive long after rank 0 published) - but only when it carries this launch's nonce (`_launch_nonce`).  An automatic
    (launcher-derived) name is accepted only while fresh - judged by the file's OWN modification time against the clock of the machine
    that reads it, both taken from the same filesystem view (no assumption that the ranks' clocks agree with the publisher's)."""
    try:
        raw = path.read_bytes()
        if len(raw) <= 128:
            return None
        stamp, _, nonce = raw[128:].partition(b"|")
        float(stamp.decode())  # well-formed payload
        if explicit_key:
            return raw[:128] if nonce == _launch_nonce(explicit_key=True) else None
        probe = path.with_name(f"{path.name}.probe{os.getpid()}")
        try:  # "now" as the filesystem that holds the file sees it (a shared directory may be served by another clock)
            fd = os.open(probe, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
            os.close(fd)
            now = os.stat(probe).st_mtime
            os.unlink(probe)
        except OSError:
            now = time.time()
        if abs(now - path.stat().st_mtime) < _fresh_seconds():
            return raw[:128]
    except (FileNotFoundError, ValueError):
        pass
    return None


def exchange_unique_id(ctx, rank: int, world: int, key: str | None = None, timeout: float = 300.0) -> bytes:
    """Rank 0 creates the communicator id and publishes it (atomic rename, replacing whatever a previous launch left under the name); the
    other ranks wait for a file of THIS launch (explicit key: its nonce; automatic name: fresh)."""
    primary, secondary = _rendezvous_files(key)
    if rank == 0:
        uid = ctx.comm_unique_id()
        for path in (primary, secondary):
            if path is not None:
                _publish(path, uid + repr(time.time()).encode() + b"|" + _launch_nonce(explicit_key=key is not None))
        return uid
    t0 = time.monotonic()
    while True:
        uid = _read_fresh(primary, explicit_key=key is not None)
        if uid is None and secondary is not None and time.monotonic() - t0 > 15.0:  # the parent-pid key found nothing: try the port key
            uid = _read_fresh(secondary, explicit_key=False)
        if uid is not None:
            return uid
        if time.monotonic() - t0 > timeout:
            raise TimeoutError(f"rank {rank}: no fresh communicator id at {primary} after {timeout:.0f} s")
        time.sleep(0.02)


def init_from_env(ctx, key: str | None = None) -> tuple[int, int]:
    """Join ``ctx`` to the RCCL communicator of the launch described by RANK / WORLD_SIZE; returns (rank, world).
    World size 1 also goes through RCCL (a one-rank communicator), so the collective code path is the one that runs."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    uid = exchange_unique_id(ctx, rank, world, key)
    ctx.comm_init(uid, rank, world)  # (librccl's version banner is kept off this process's stdout, see Context.comm_init)
    ctx.comm_barrier()
    if rank == 0:  # everyone has read the id once the barrier returns
        for path in _rendezvous_files(key):
            try:
                if path is not None:
                    path.unlink()
            except OSError:
                pass
    return rank, world

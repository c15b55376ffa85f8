"""``DistributedContractor`` for the hip backend (reference ``tensorcircuit/experimental.py:760-1249``).

Same constructor / methods / ``tree_data`` format as the reference.  MI355X-first differences:

* one process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI) instead of one JAX
  process driving all devices; without an initialised process group the single rank runs every slice;
* the path search is shared: the seeds of the hyper-search are dealt to the ranks and the best tree is
  broadcast (``_get_tree_data``; the reference's rank-0 search + ``broadcast_py_object``,
  ``experimental.py:850-857``); the search is deterministic per seed, so every world size picks the same tree;
* slices are contracted by the HIP tensordot engine (``tcmi/tn.py``) and the per-rank partial
  ``[value || flattened gradients]`` is summed with ONE packed all-reduce
  (reference ``jnp.sum(device_values, axis=0)``, ``experimental.py:1145-1152``).
"""

import itertools
import pickle
from typing import Any, Callable, Dict, List, Optional

import numpy as np

from . import cons
from .circuit import upload_cached
from . import distributed as D
from . import tn

Tensor = Any
_TRACE_SERIAL = itertools.count(1)      # one number per recorded node-function trace, process-wide


_SEARCH_DIGEST: List[bytes] = []


def _search_one_seed(job):
    """One seed of the hyper-search (module level: also the worker of the process pool of ``parallel``): random-greedy
    trials, slicing, reconfiguration; returns (objective, position in the seed list, statistics, tree data)."""
    import time as _time

    inputs, output, size_dict, o, pos, seed, rank = job
    t0 = _time.perf_counter()
    tree = tn.ContractionTree.from_path(inputs, output, size_dict, trials=o["max_repeats"], seed=seed)
    tree.minimize = o["minimize"]
    if o["target_size"] is not None:
        tree.slice_to(o["target_size"])
    if o["target_slices"] is not None:
        tree.slice_to_slices(o["target_slices"])
    obj = tree.objective()
    key = tuple(float(x) for x in (obj if isinstance(obj, tuple) else (obj,)))
    stat = {"seed": seed, "objective": list(key), "model_time_s": tree.model_time(), "nslices": int(tree.nslices),
            "search_s": round(_time.perf_counter() - t0, 2), "rank": rank}
    return key, pos, stat, tree.to_data()


def _search_seeds_in_subprocesses(jobs, nproc: int):
    """``_search_one_seed`` of every job in fresh interpreter processes, ``nproc`` at a time (plain subprocesses fed through
    pipes, not a multiprocessing pool: a pool's children re-import the caller's main module, which a library must not
    assume is import-safe).  Falls back to None (the caller searches serially) if a child fails."""
    import subprocess
    import sys

    code = ("import sys, pickle\n"
            "sys.path[:0] = pickle.loads(bytes.fromhex(sys.argv[1]))\n"
            "from tcmi.experimental import _search_one_seed\n"
            "job = pickle.load(sys.stdin.buffer)\n"
            "sys.stdout.buffer.write(pickle.dumps(_search_one_seed(job)))\n")
    paths = pickle.dumps([p for p in sys.path if p]).hex()
    out: List[Any] = [None] * len(jobs)
    pending = list(enumerate(jobs))
    running: List[Any] = []
    try:
        while pending or running:
            while pending and len(running) < max(1, nproc):
                i, job = pending.pop(0)
                pr = subprocess.Popen([sys.executable, "-c", code, paths], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                      stderr=subprocess.DEVNULL)
                pr.stdin.write(pickle.dumps(job))
                pr.stdin.close()
                running.append((i, pr))
            i, pr = running.pop(0)
            data = pr.stdout.read()
            if pr.wait() != 0:
                raise RuntimeError("seed worker failed")
            out[i] = pickle.loads(data)
        return out
    except Exception:  # noqa: BLE001
        for _, pr in running:
            pr.kill()
        return None


def _tree_cache_key(inputs, output, size_dict, opts) -> Optional[str]:
    """Digest of everything a searched tree depends on: the renumbered index structure, the parsed options, and the text
    of the search itself (tcmi/tn.py + the native helpers' version: any change of the algorithm empties the cache)."""
    import hashlib
    import os

    if os.environ.get("TCMI_TREE_CACHE", "1") == "0":
        return None
    if not _SEARCH_DIGEST:
        h = hashlib.blake2b(digest_size=16)
        with open(tn.__file__, "rb") as fh:
            h.update(fh.read())
        from . import specialize as _S

        for f in ("tcmi_host.cpp",):
            try:
                with open(os.path.join(_S.CSRC, f), "rb") as fh:
                    h.update(fh.read())
            except OSError:
                pass
        _SEARCH_DIGEST.append(h.digest())
    h = hashlib.blake2b(digest_size=16)
    h.update(_SEARCH_DIGEST[0])
    h.update(repr((inputs, output, sorted(size_dict.items()), sorted((k, v) for k, v in opts.items() if k != "parallel"))).encode())
    return h.hexdigest()


def _tree_cache_dirs() -> List[str]:
    import os

    from . import specialize as _S

    return [os.path.join(d, "trees") for d in _S._cache_dirs()]


def _tree_cache_load(key: Optional[str]):
    import os

    if key is None:
        return None
    for d in _tree_cache_dirs():
        p = os.path.join(d, key + ".pkl")
        if os.path.exists(p):
            try:
                with open(p, "rb") as fh:
                    rec = pickle.load(fh)
                if isinstance(rec, dict) and "data" in rec and "stats" in rec:
                    return rec
            except Exception:  # noqa: BLE001  (a truncated file: search again)
                pass
    return None


def _tree_cache_store(key: Optional[str], data, stats) -> None:
    import os

    if key is None:
        return
    for d in _tree_cache_dirs():
        try:
            os.makedirs(d, exist_ok=True)
            tmp = os.path.join(d, f"{key}.{os.getpid()}.tmp")
            with open(tmp, "wb") as fh:
                pickle.dump({"data": data, "stats": [dict(st) for st in stats]}, fh)
            os.replace(tmp, os.path.join(d, key + ".pkl"))       # atomic: a reader sees the whole file or none
            return
        except OSError:
            continue


class DistributedContractor:
    last_search: List[Dict[str, Any]] = []     # per-seed record of the most recent path search (bench.py reports it)

    def __init__(self, nodes_fn: Callable[[Any], List[tn.Node]], params: Any,
                 cotengra_options: Optional[Dict[str, Any]] = None, devices: Optional[List[Any]] = None,
                 mesh: Optional[Any] = None, tree_data: Optional[Dict[str, Any]] = None) -> None:
        import torch.distributed as dist

        self.nodes_fn = nodes_fn
        self._backend = "hip"
        self.rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
        self.num_devices = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        if tree_data is None:
            tree_data = self._get_tree_data(nodes_fn, params, cotengra_options)
        self.tree = tn.ContractionTree.from_data(tree_data)
        self._report_tree_info()
        # reference experimental.py:881-890: [num_devices, ceil(S / num_devices)], padded with -1
        self.slice_table = D.slice_table(self.tree.nslices, self.num_devices)
        self.my_slices = [int(s) for s in self.slice_table[self.rank] if s >= 0]

    # ---- path search / persistence (reference experimental.py:923-991) --------------------------
    # cotengra options this backend understands (reference experimental.py:934-946 hands the dictionary to
    # ``ctg.ReusableHyperOptimizer(**opts)``); anything else raises instead of being dropped
    _SLICING_KEYS = ("target_size", "target_slices")

    @staticmethod
    def _parse_options(cotengra_options: Optional[Dict[str, Any]]) -> Dict[str, Any]:
        opts = dict(cotengra_options or {})
        out: Dict[str, Any] = {"target_size": None, "target_slices": None, "minimize": None}
        out["max_repeats"] = int(opts.pop("max_repeats", 128))      # reference default (experimental.py:936-942)
        sd = opts.pop("seed", 0)        # one seed, or several: a small hyper-search, the best tree by the objective wins
        out["seeds"] = [int(x) for x in sd] if isinstance(sd, (list, tuple, range)) else [int(sd)]
        for key in ("slicing_opts", "slicing_reconf_opts"):
            sub = dict(opts.pop(key, None) or {})
            for k in DistributedContractor._SLICING_KEYS:
                if k in sub:
                    out[k] = int(sub.pop(k))
            sub.pop("max_repeats", None)         # repeats of the reconfiguration: the beam of tn._reconfigure_sliced
            if sub:
                raise NotImplementedError(f"Backend 'hip' has not implemented cotengra {key} keys {sorted(sub)} "
                                          f"(understood: {DistributedContractor._SLICING_KEYS})")
        if "minimize" in opts:
            m = opts.pop("minimize")
            if not isinstance(m, str) or not (m in ("flops", "write", "size", "max") or m.startswith("combo")):
                raise NotImplementedError(f"Backend 'hip' has not implemented minimize={m!r}")
            out["minimize"] = m
        methods = opts.pop("methods", None)
        if methods is not None:
            methods = [methods] if isinstance(methods, str) else list(methods)
            bad = [m for m in methods if m not in ("greedy", "random-greedy")]
            if bad:
                raise NotImplementedError(f"Backend 'hip' has not implemented the path-search methods {bad}: the search is "
                                          f"random-greedy + subtree reconfiguration (methods=['greedy'])")
        # execution hints without an effect on the result: every rank derives the same tree in-process (deterministic
        # search, native subtree programme), so there is nothing to parallelise over and nothing to show
        # ``parallel`` (cotengra: search trials on a process pool): the SEEDS of the hyper-search on a process pool, where a
        # pool can be started (no GPU context in this process yet: build hosts, offline find_path); ``progbar``: nothing to show
        par = opts.pop("parallel", False)
        out["parallel"] = 0 if not par else (int(par) if not isinstance(par, bool) else -1)
        opts.pop("progbar", None)
        if opts:
            raise NotImplementedError(f"Backend 'hip' has not implemented the cotengra options {sorted(opts)}")
        if out["target_size"] is None and out["target_slices"] is None:
            out["target_size"] = 2**28           # reference default slicing_reconf_opts (experimental.py:936-942)
        return out

    @staticmethod
    def _target_size(cotengra_options: Optional[Dict[str, Any]]) -> int:
        ts = DistributedContractor._parse_options(cotengra_options)["target_size"]
        return 2**28 if ts is None else ts

    @staticmethod
    def _get_tree_data(nodes_fn, params, cotengra_options, collective: bool = True) -> Dict[str, Any]:
        o = DistributedContractor._parse_options(cotengra_options)
        nodes = nodes_fn(params)
        inputs, output, size_dict = tn.get_tn_info(nodes)
        # edge labels are renumbered 0..E-1 in first-appearance order: valid for every re-trace
        ren: Dict[int, int] = {}
        for s in inputs:
            for e in s:
                ren.setdefault(e, len(ren))
        inputs = [[ren[e] for e in s] for s in inputs]
        output = [ren[e] for e in output]
        size_dict = {ren[e]: d for e, d in size_dict.items()}
        # The search (random-greedy trials, slicing, subtree reconfiguration) is deterministic per seed but its result
        # varies a lot from seed to seed (32-qubit RQC: 19 to 90 ms of model time over seeds 0..7): with several seeds the
        # whole pipeline runs once per seed and the best tree by the objective (``minimize``; default: the engine's
        # two-roof time model) is kept -- the role of cotengra's hyper-optimiser (reference experimental.py:934-953).
        import os
        import time as _time

        import torch.distributed as dist

        # ``collective=False`` (find_path: typically called by ONE process): no collective, the whole search locally
        world = dist.get_world_size() if (collective and dist.is_available() and dist.is_initialized()) else 1
        rank = dist.get_rank() if world > 1 else 0
        # Searched trees are kept (the role of cotengra's ReusableHyperOptimizer, which the reference builds its optimiser
        # with, experimental.py:934-953: paths cached by network): a tree depends on the network's index structure and
        # the options only, the search is deterministic, so a second contractor of the same network -- the next run of the
        # same program -- loads it from the cache directory of the generated kernels instead of searching for seconds
        # again.  TCMI_TREE_CACHE=0 switches it off.  With several ranks, rank 0 looks it up and broadcasts hit or miss.
        ckey = _tree_cache_key(inputs, output, size_dict, o)
        hit = _tree_cache_load(ckey) if rank == 0 else None
        if world > 1:
            box = [hit]
            dist.broadcast_object_list(box, src=0)
            hit = box[0]
        if hit is not None:
            DistributedContractor.last_search = [dict(st, cached=True) for st in hit["stats"]]
            return hit["data"]
        # With an initialised process group the SEEDS are dealt to the ranks (seed k to rank k mod W): every rank searches
        # its share, one all_gather_object of the objectives picks the winner (smallest objective, ties to the seed that
        # comes first in the list: what the serial loop keeps) and the winner's rank broadcasts the tree -- with one seed
        # this is the reference's rank-0 search + ``broadcast_py_object`` (experimental.py:850-857), with 8 seeds on 8
        # ranks the Python-bound search takes the time of one seed instead of eight on every rank.  TCMI_TN_SEARCH_SHARD=0:
        # every rank runs the whole deterministic search locally (no collective in the constructor).
        shard = world > 1 and os.environ.get("TCMI_TN_SEARCH_SHARD", "1") != "0"
        seeds = list(enumerate(o["seeds"]))
        mine = seeds[rank::world] if shard else seeds
        jobs = [(inputs, output, size_dict, {k: o[k] for k in ("max_repeats", "minimize", "target_size", "target_slices")},
                 pos, seed, rank) for pos, seed in mine]
        results = None
        if o.get("parallel") and world == 1 and len(jobs) > 1:
            import torch

            if not torch.cuda.is_initialized():          # (a process that holds a GPU context starts no children here)
                nproc = min(len(jobs), (os.cpu_count() or 1) if o["parallel"] < 0 else o["parallel"])
                results = _search_seeds_in_subprocesses(jobs, nproc)
        if results is None:
            results = [_search_one_seed(j) for j in jobs]
        best, stats = None, []
        for key, pos, stat, data_ in results:
            stats.append(stat)
            if best is None or (key, pos) < best[0]:
                best = ((key, pos), data_)
        if not shard:
            DistributedContractor.last_search = stats
            data = best[1]
            if rank == 0:
                _tree_cache_store(ckey, data, stats)
            return data
        gathered = [None] * world
        dist.all_gather_object(gathered, (None if best is None else best[0], stats))
        cands = [(kp, r) for r, (kp, _) in enumerate(gathered) if kp is not None]
        winner = min(cands)[1]
        DistributedContractor.last_search = sorted((st for _, sts in gathered for st in sts),
                                                   key=lambda st: o["seeds"].index(st["seed"]))
        box = [best[1] if rank == winner else None]
        dist.broadcast_object_list(box, src=winner)
        if rank == 0:
            _tree_cache_store(ckey, box[0], DistributedContractor.last_search)
        return box[0]

    @staticmethod
    def find_path(nodes_fn: Callable[[Any], List[tn.Node]], params: Any,
                  cotengra_options: Optional[Dict[str, Any]] = None, filepath: Optional[str] = None) -> Dict[str, Any]:
        data = DistributedContractor._get_tree_data(nodes_fn, params, cotengra_options, collective=False)
        if filepath is not None:
            with open(filepath, "wb") as f:
                pickle.dump(data, f)
        return data

    @classmethod
    def from_path(cls, filepath: str, nodes_fn: Callable[[Any], List[tn.Node]], devices: Optional[List[Any]] = None,
                  mesh: Optional[Any] = None, params: Any = None) -> "DistributedContractor":
        with open(filepath, "rb") as f:
            data = pickle.load(f)
        return cls(nodes_fn, params, devices=devices, mesh=mesh, tree_data=data)

    def _report_tree_info(self) -> None:
        """reference experimental.py:909-920."""
        t = self.tree
        item = 8 if cons.dtypestr == "complex64" else 16
        self.tree_info = {
            "nslices": t.nslices, "sliced_inds": len(t.sliced_inds), "log2_max_size": t.contraction_width(),
            "log10_flops": float(np.log10(max(1, t.total_flops()))), "log2_write": float(np.log2(max(1, t.total_write()))),
            "algorithmic_bytes": t.algorithmic_bytes(item),
        }

    # ---- evaluation -----------------------------------------------------------------------------------
    def _arrays(self, params):
        """The leaf tensors of the network for ``params``.  The node function is TRACED (the reference jits it,
        experimental.py:1182-1211): the first call records how every gate tensor came out of the batched stacks of
        ``Circuit._gate_stacks`` (constants, or ``C0 + cos C1 + sin C2`` of angles gathered from one parameter tensor), the
        second call checks that the recorded recipe reproduces the function's tensors bit for bit, and from then on only
        the recipe runs -- a handful of batched torch ops instead of one Python gate call per gate (20 -> 3 ms per call
        at 944 gates).  Anything the recipe cannot express (a gate tensor that is on the tape but did not come out of a
        stack, angles that are not elements of one parameter tensor) keeps the function.  TCMI_TN_TRACE=0 disables."""
        import os

        import torch

        from . import circuit as _circ

        st = getattr(self, "_trace_state", None)
        if st is None:
            st = self._trace_state = {"mode": "off" if os.environ.get("TCMI_TN_TRACE", "1") == "0" else "record", "recipe": None}
        leaves = [x for x in cons.backend.tree_flatten(params)[0]] if params is not None else []
        if st["mode"] == "replay":
            out = self._replay_recipe(st["recipe"], leaves)
            if out is not None:
                return out
            st["mode"] = "off"
        if st["mode"] == "off":
            return [n.tensor for n in self.nodes_fn(params)]
        rec = _circ.NodeTrace()
        keep, _circ.NODE_TRACE = _circ.NODE_TRACE, rec
        try:
            nodes = self.nodes_fn(params)
        finally:
            _circ.NODE_TRACE = keep
        arrays = [n.tensor for n in nodes]
        recipe = self._build_recipe(rec, arrays, leaves)
        if recipe is None:
            st["mode"] = "off"
        elif st["recipe"] is None:
            st["recipe"] = recipe            # first call: remember, check against the function on the next one
            # what later calls cache per recipe (value_and_grad's fast_key) is keyed by this serial, never by id(recipe):
            # an id can come back after a re-trace has freed the old recipe
            st["serial"] = next(_TRACE_SERIAL)
            self._fast_vjp_ok = self._group_cache = None
        else:
            again = self._replay_recipe(st["recipe"], leaves)
            same = again is not None and len(again) == len(arrays) and all(
                a.shape == b.shape and a.dtype == b.dtype and a.device == b.device for a, b in zip(again, arrays))
            if same:      # ONE comparison of everything (a torch.equal per tensor is a device round trip each)
                by_dt: Dict[Any, list] = {}
                for a, b in zip(again, arrays):
                    by_dt.setdefault(a.dtype, []).append((a.detach().reshape(-1), b.detach().reshape(-1)))
                same = all(torch.equal(torch.cat([x for x, _ in prs]), torch.cat([y for _, y in prs])) for prs in by_dt.values())
            if same and leaves:
                same = self._constants_ignore_the_arguments(params, arrays, st["recipe"])
            st["mode"] = "replay" if same else "off"
        return arrays

    def _constants_ignore_the_arguments(self, params, arrays, recipe) -> bool:
        """The recipe freezes every node tensor that did not come out of a gate stack as a CONSTANT of the network.  That
        is only right if those tensors do not depend on the arguments -- the one-hot caps of
        ``amplitude_before(params["bitstring"])``, ``Circuit(inputs=f(params))`` or a matrix baked from an argument's
        value do, and a replay would keep returning the first call's network (the reference jits nodes_fn: every argument
        stays live there).  Checked once, on the validating call: the node function is run on PERTURBED arguments (every
        element of every leaf changed) and the would-be constants must come out unchanged; a function that cannot even
        run on them (a bitstring that is no longer one) is not traced either."""
        import torch

        K = cons.backend
        try:
            leaves, spec = K.tree_flatten(params)
            moved = []
            for x in leaves:
                if not torch.is_tensor(x):
                    x = K.convert_to_tensor(x)
                x = x.detach()
                if x.is_floating_point() or x.is_complex():
                    moved.append(x + 0.7311)
                elif x.dtype == torch.bool:
                    moved.append(~x)
                else:
                    moved.append(torch.where((x == 0) | (x == 1), 1 - x, x + 1))
            with torch.no_grad():
                other = [n.tensor for n in self.nodes_fn(K.tree_unflatten(spec, moved))]
            if len(other) != len(arrays):
                return False
            for a, b, it in zip(arrays, other, recipe["items"]):
                if not torch.is_tensor(it):
                    continue
                if a.shape != b.shape or a.dtype != b.dtype or not torch.equal(a.detach(), b.detach()):
                    return False
            return True
        except Exception:  # noqa: BLE001
            return False

    @staticmethod
    def _build_recipe(rec, arrays, leaves):
        import torch

        stacks = []
        for ent in rec.stacks:
            if ent[0] == "const":
                stacks.append(("const", ent[1]))
            elif ent[0] == "trig":
                li = next((i for i, x in enumerate(leaves) if x is ent[4]), None)
                if li is None:
                    return None
                stacks.append(("trig", ent[1], ent[2], ent[3], li, tuple(ent[4].shape), ent[4].dtype))
            else:
                return None
        items = []
        for t in arrays:
            src = getattr(t, "_tcmi_src", None)
            if src is not None:
                items.append((src[0], src[1], src[2], tuple(t.shape)))
            elif torch.is_tensor(t) and not t.requires_grad and not torch._C._functorch.is_functorch_wrapped_tensor(t):
                items.append(t)                 # a constant of the network (|0>, the measured operator)
            else:
                return None
        return {"stacks": stacks, "items": items}

    @staticmethod
    def _replay_recipe(recipe, leaves):
        import torch

        from .circuit import trig_stack

        plain, conj = [], {}
        for ent in recipe["stacks"]:
            if ent[0] == "const":
                plain.append(ent[1])
                continue
            _, cdev, aff, offs, li, shape, dtype = ent
            if li >= len(leaves) or not torch.is_tensor(leaves[li]) or tuple(leaves[li].shape) != shape or \
                    leaves[li].dtype != dtype or leaves[li].device != cdev.device or not leaves[li].is_contiguous():
                return None
            plain.append(trig_stack(cdev, aff, leaves[li].reshape(-1)[offs]))
        # the gate tensors of one stack that share a shape come out of ONE unbind of the stack viewed as [rows, *shape]
        # (a thousand ``stk[row].reshape(shape)`` calls were 2 ms of host time per call of a 30-qubit depth-8 ladder);
        # every view's _base is still the 2-D stack, which is what value_and_grad groups the cotangents by
        rows_of: Dict[Any, Any] = {}
        out = []
        for it in recipe["items"]:
            if torch.is_tensor(it):
                out.append(it)
                continue
            sid, row, cj, shape = it
            key = (sid, cj, shape)
            views = rows_of.get(key)
            if views is None:
                stk = plain[sid]
                if cj:
                    if sid not in conj:
                        conj[sid] = stk.conj().resolve_conj()
                    stk = conj[sid]
                views = rows_of[key] = stk.reshape((stk.shape[0],) + tuple(shape)).unbind(0)
            out.append(views[row])
        return out

    def _shard(self):
        """(rank, world, group) for the split of the slice-invariant subtrees (TCMI_TN_SHARD_INV=0: every rank computes
        all of them, as the reference's pmap replicas do)."""
        import os

        emu = getattr(self, "_emulate_rank", None)
        if emu is not None:       # bench.py: ONE process executes what rank emu[0] of emu[1] would (no collectives; the
            return (emu[0], emu[1], "emulate")   # other ranks' invariant roots are computed here once, outside the timing)
        if self.num_devices > 1 and os.environ.get("TCMI_TN_SHARD_INV", "1") != "0":
            return (self.rank, self.num_devices, None)
        return None

    def _local_sum(self, params, op):
        total = None
        arrays = self._arrays(params)
        for r in self.tree.contract_slices(arrays, self.my_slices, shard=self._shard()):
            total = r if total is None else total + r
        if total is None:  # this rank only holds padding
            import torch

            total = torch.zeros([2] * (len(self.tree.output)), dtype=getattr(torch, cons.dtypestr), device=cons.backend.device)
        return total

    def value(self, params: Any, op: Optional[Callable[[Tensor], Tensor]] = None, output_dtype: Optional[str] = None) -> Tensor:
        """Sum over all slices and ranks; default ``op`` = sum of the result (complex), reference
        experimental.py:1065-1100."""
        import torch

        with torch.no_grad():
            part = self._local_sum(params, op)
            re, im = D.allreduce_sum_packed([part.real.contiguous(), part.imag.contiguous()])
            full = torch.complex(re, im)
        out = op(full) if op is not None else full.sum()
        if output_dtype is not None:
            out = cons.backend.cast(out, output_dtype)
        return out

    def value_and_grad(self, params: Any, op: Optional[Callable[[Tensor], Tensor]] = None, output_dtype: Optional[str] = None):
        """(value, grads) with grads shaped like ``params``; default op = real(sum(x))
        (reference experimental.py:1018,1182-1211).  The op must be linear in the contraction result
        for the slice sum to commute with it (as in the reference, which applies op per slice)."""
        import torch

        K = cons.backend
        leaves, spec = K.tree_flatten(params)
        leaves = [K.convert_to_tensor(x).detach().clone().requires_grad_(True) for x in leaves]
        p = K.tree_unflatten(spec, leaves)
        fop = op if op is not None else (lambda x: x.sum().real)
        arrays = self._arrays(p)
        # A traced node function replays ONE recipe: the arrays of every call have the same count, shapes, dtypes and tape
        # membership, so what was established about them once (the hand-written sweep takes them, the graphs fit them, which
        # rows of which stack they are) is not re-derived array by array on every call -- with a thousand gate tensors those
        # loops were most of the host time of a step (scripts/gpu_svqa_host.py).
        st_ = getattr(self, "_trace_state", None)
        fast_key = None
        if st_ is not None and st_["mode"] == "replay":
            fast_key = (("trace", st_["serial"]), tuple((tuple(x.shape), x.dtype) for x in leaves), cons.dtypestr)
        fk = getattr(self, "_fast_vjp_ok", None)
        if fast_key is not None and fk == fast_key:
            fast = True
        else:
            fast = self._fast_vjp(arrays)
            self._fast_vjp_ok = fast_key if fast else None
        if fast:
            # reverse sweep over the step list on the untaped kernels (tn.contract_slices_vjp); the small gate tensors
            # stay on torch's tape, so their cotangents reach ``params`` through one autograd call
            value, agrads = self.tree.contract_slices_vjp(arrays, self.my_slices, fop, alias_ok=True, hat_ok=True,
                                                          shard=self._shard(), fast_key=fast_key)
            hat = bool(getattr(self.tree, "last_vjp_conjugated", False))   # the sweep handed over conj(g): undone per stack
            if value is None:
                value = sum((x.sum() * 0 for x in leaves)).real.detach()
            # (which arrays receive a cotangent depends on the slices and the invariant subtrees this rank holds)
            sh_ = self._shard()
            gkey = None if fast_key is None else (fast_key, tuple(self.my_slices),
                                                  None if sh_ is None else (sh_[0], sh_[1], sh_[2] == "emulate"))
            grouped = self._grouped_cotangents(gkey, arrays, agrads, hat) if gkey is not None else None
            if grouped is not None:
                outs, gouts = grouped
                grads = torch.autograd.grad(outs, leaves, gouts, allow_unused=True) if outs else [None] * len(leaves)
                pairs = None
            else:
                pairs = [(a, g) for a, g in zip(arrays, agrads) if g is not None and a.requires_grad]
            if grouped is not None:
                pass
            elif pairs:
                # gate tensors are rows of a few stacks (Circuit._gate_stacks): hand autograd one cotangent per STACK --
                # a thousand row selections would each run their own backward node
                outs, gouts, groups = [], [], {}
                for a, g in pairs:
                    b = a._base
                    if b is not None and b.dim() == 2 and b.requires_grad and b.is_contiguous() and b.storage_offset() == 0 \
                            and a.is_contiguous() and a.numel() == b.shape[1] and a.storage_offset() % b.shape[1] == 0:
                        groups.setdefault(id(b), (b, [], []))
                        groups[id(b)][1].append(a.storage_offset() // b.shape[1])
                        groups[id(b)][2].append(g.reshape(-1))
                    else:
                        outs.append(a)
                        gouts.append(g.conj().resolve_conj() if hat else g)
                for b, rows, gs in groups.values():
                    gb = torch.zeros_like(b)
                    st = torch.stack(gs)
                    gb.index_add_(0, upload_cached(np.asarray(rows, dtype=np.int64), None, b.device),
                                  st.conj().resolve_conj() if hat else st)
                    outs.append(b)
                    gouts.append(gb)
                grads = torch.autograd.grad(outs, leaves, gouts, allow_unused=True)
            else:
                grads = [None] * len(leaves)
        else:
            value = None
            for r in self.tree.contract_slices(arrays, self.my_slices):
                r = fop(r)
                value = r if value is None else value + r
            if value is None:
                value = sum((x.sum() * 0 for x in leaves)).real
            grads = torch.autograd.grad(value, leaves, allow_unused=True)
        grads = [g if g is not None else torch.zeros_like(x) for g, x in zip(grads, leaves)]
        packed = D.allreduce_sum_packed([value.detach().reshape(1).real.to(torch.float64)] + [g.to(torch.float64) if not g.is_complex() else g.real.to(torch.float64) for g in grads])
        v = packed[0].reshape(())
        gs = [g.to(x.dtype) for g, x in zip(packed[1:], leaves)]
        if output_dtype is not None:
            v = K.cast(v, output_dtype)
        else:
            v = v.to(getattr(torch, cons.rdtypestr))
        return v, K.tree_unflatten(spec, gs)

    def _grouped_cotangents(self, fast_key, arrays, agrads, hat):
        """(stack tensors, their cotangents) for ``torch.autograd.grad``: the rows-of-stacks grouping of value_and_grad with
        its structure (which array is which row of which stack) cached per recipe.  None: no cached structure applies (the
        caller groups array by array and, if every array on the tape is a stack row, the structure is remembered)."""
        import torch

        gc = getattr(self, "_group_cache", None)
        if gc is None or gc["key"] != fast_key:
            groups, order = {}, []
            for k, (a, g) in enumerate(zip(arrays, agrads)):
                if g is None or not a.requires_grad:
                    continue
                b = a._base
                if not (b is not None and b.dim() == 2 and b.requires_grad and b.is_contiguous() and b.storage_offset() == 0
                        and a.is_contiguous() and a.numel() == b.shape[1] and a.storage_offset() % b.shape[1] == 0):
                    return None             # an array on the tape that is not a stack row: the general route
                if id(b) not in groups:
                    groups[id(b)] = {"k0": k, "rows": [], "ks": []}
                    order.append(id(b))
                groups[id(b)]["rows"].append(a.storage_offset() // b.shape[1])
                groups[id(b)]["ks"].append(k)
            gl = []
            for bid in order:
                gr = groups[bid]
                gl.append((gr["k0"], upload_cached(np.asarray(gr["rows"], dtype=np.int64), None, arrays[gr["k0"]].device), gr["ks"]))
            gc = self._group_cache = {"key": fast_key, "groups": gl}
        outs, gouts = [], []
        for k0, rows, ks in gc["groups"]:
            b = arrays[k0]._base
            st = torch.stack([agrads[k].reshape(-1) for k in ks])
            gb = torch.zeros_like(b)
            gb.index_add_(0, rows, st.conj().resolve_conj() if hat else st)
            outs.append(b)
            gouts.append(gb)
        return outs, gouts

    @staticmethod
    def _fast_vjp(arrays) -> bool:
        """The hand-written reverse sweep needs plain complex [2]^rank device tensors (TCMI_TN_VJP=0: torch's tape)."""
        import os
        import torch

        if os.environ.get("TCMI_TN_VJP", "1") == "0":
            return False
        return all(torch.is_tensor(t) and t.is_cuda and t.is_complex() and all(d == 2 for d in t.shape)
                   and not torch._C._functorch.is_functorch_wrapped_tensor(t) for t in arrays)

    def grad(self, params: Any, op: Optional[Callable[[Tensor], Tensor]] = None, output_dtype: Optional[str] = None) -> Any:
        return self.value_and_grad(params, op, output_dtype)[1]


# ---- output-wavefunction slicing (reference examples/slicing_wavefunction_vqa.py:20-36,76-103) -------------------------
def sliced_state(c, cut: List[int], mask) -> Tensor:
    """The output state of circuit ``c`` projected on ``|mask>`` of the qubits ``cut``: 2^(n - len(cut)) amplitudes with
    the remaining qubits in circuit order, contracted as a tensor network with the projected legs capped -- the whole
    2^n state never exists (reference ``sliced_state``: one-hot ends on ``front[cut]``, ``tc.contractor(nodes + ends,
    output_edge_order=...)``).  Differentiable in the gate parameters (``tn.TensordotFn``)."""
    import torch

    from . import tn as _tn

    n = c._nqubits
    cut = [int(q) % n for q in cut]
    if len(set(cut)) != len(cut):
        raise ValueError("cut qubits must be distinct")
    mask = [int(round(float(m))) for m in (cons.backend.numpy(mask).reshape(-1) if not isinstance(mask, (list, tuple)) else mask)]
    if len(mask) != len(cut):
        raise ValueError("one mask bit per cut qubit")
    dt = getattr(torch, cons.dtypestr)
    nodes, front = c._tn_nodes()
    for q, b in zip(cut, mask):
        v = torch.zeros(2, dtype=dt, device=cons.backend.device)
        v[b] = 1.0
        nodes.append(_tn.Node(v, [front[q]], "slice-end"))
    oeo = [front[i] for i in range(n) if i not in cut]
    res = _tn.contract_nodes(nodes, output_edge_order=oeo)
    return res.tensor.reshape(-1)


def sliced_expectation_ps(circuit_fn: Callable[[], Any], ps, cut: List[int], masks=None) -> Tensor:
    """<psi| P |psi> of the Pauli string ``ps`` (0/1/2/3 per qubit) as a sum over the 2^k projections of the ``cut``
    qubits, each term built from two sliced states of 2^(n-k) amplitudes (reference ``sliced_core`` /
    ``sliced_expectation_and_grad``): P on the cut qubits maps |m> to phase(m) |m ^ x_cut>, the rest of the string is
    applied to the sliced ket (module-level ``expectation(ket=, bra=)``).  The 2^k terms are independent: with an
    initialised process group they are dealt to the ranks in contiguous blocks (``distributed.shard_range``) and the
    partial sums meet in one all-reduce -- the multi-GPU route for a state that does not fit one HBM.  ``masks``
    restricts the sum (testing).  Returns the real part (``ps`` is Hermitian)."""
    import itertools

    import torch
    import torch.distributed as dist

    from . import gates as _G
    from .circuit import expectation as state_expectation

    c = circuit_fn()
    n = c._nqubits
    ps = [int(p) for p in ps]
    cut = [int(q) % n for q in cut]
    rest = [i for i in range(n) if i not in cut]
    pos = {q: j for j, q in enumerate(rest)}
    ops_rest = [(getattr(_G, "ixyz"[ps[q]])(), [pos[q]]) for q in rest if ps[q] != 0]
    allm = list(itertools.product((0, 1), repeat=len(cut))) if masks is None else [tuple(m) for m in masks]
    rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    lo, hi = D.shard_range(len(allm), rank, world)
    total = None
    for m1 in allm[lo:hi]:
        # P_cut |m1> = phase |m2>: X, Y flip the bit; Y contributes i (-1)^b, Z contributes (-1)^b
        phase, m2 = 1.0 + 0.0j, list(m1)
        for j, q in enumerate(cut):
            b = m1[j]
            if ps[q] == 1:
                m2[j] = 1 - b
            elif ps[q] == 2:
                m2[j] = 1 - b
                phase *= 1j * (1 - 2 * b)
            elif ps[q] == 3:
                phase *= (1 - 2 * b)
        ket = sliced_state(circuit_fn(), cut, m1)
        bra = ket if tuple(m2) == tuple(m1) else sliced_state(circuit_fn(), cut, m2)
        term = phase * state_expectation(*ops_rest, ket=ket, bra=bra)
        total = term if total is None else total + term
    if total is None:
        total = torch.zeros((), dtype=getattr(torch, cons.dtypestr), device=cons.backend.device)
    if world > 1:
        re, im = D.allreduce_sum_packed([total.real.reshape(1), total.imag.reshape(1)])
        return re.reshape(())
    return total.real

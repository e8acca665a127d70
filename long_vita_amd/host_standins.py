"""Stand-ins for the Ascend-only modules the reference imports on its attention path, for a host without ``torch_npu``.

The reference's files import, at module level,
  mindspeed.utils                  get/set_actual_seq_len, get/set_position_ids, compute_actual_seq_len
                                   (lcvlm_modellink/training/utils.py:29, dot_product_attention.py:25,
                                   rotary_pos_embedding.py:18)
  mindspeed.patch_utils            MindSpeedPatchesManager (lcvlm_modellink/megatron_adaptor.py:108)
  mindspeed.core.context_parallel.ring_context_parallel   ringattn_context_parallel (dot_product_attention.py:12)
and MindSpeed itself imports ``torch_npu``, which does not exist for AMD GPUs.  ``install()`` puts small modules of
those names into ``sys.modules`` (only the names that are not importable already), backed by this package:
the accessors are the ones in ``long_vita_amd.state``, the ring is ``long_vita_amd.ring_context_parallel`` and the
patch manager below applies a patch by importing the dotted module path and setting the attribute.

Nothing here is needed by this package itself (tests and the benchmark set ``long_vita_amd.state`` directly)."""
from __future__ import annotations

import importlib
import sys
import types

from . import state


class PatchesManager:
    """``register_patch(dotted_name, fn, force_patch=False)`` then ``apply_patches()``, the two calls the reference
    makes (lcvlm_modellink/megatron_adaptor.py:11-13,107-118).  A name registered twice keeps the first function
    unless the later call forces."""

    patches_info: dict = {}

    @classmethod
    def register_patch(cls, orig_func_name, new_func=None, force_patch=False, create_dummy=False):
        if orig_func_name in cls.patches_info and not force_patch:
            return
        cls.patches_info[orig_func_name] = new_func

    @classmethod
    def apply_patches(cls):
        applied = []
        for dotted, fn in cls.patches_info.items():
            if fn is None:
                continue
            parts = dotted.split(".")
            # the longest importable prefix is the module; the rest is an attribute path (a class attribute, say)
            for cut in range(len(parts) - 1, 0, -1):
                try:
                    target = importlib.import_module(".".join(parts[:cut]))
                except ImportError:
                    continue
                for name in parts[cut:-1]:
                    target = getattr(target, name)
                setattr(target, parts[-1], fn)
                applied.append(dotted)
                break
            else:
                raise ImportError(f"cannot apply patch: no importable module in '{dotted}'")
        return applied


def compute_actual_seq_len(position_ids):
    """Cumulative END offsets of the packed segments from per-segment-restarting position ids ([b, s] or [s]): a new
    segment starts wherever the id is 0.  (MindSpeed's function of this name is not in the reference tree; the
    reference's data path sets the list itself, training/utils.py:670-672, so this is the fallback only.)"""
    flat = position_ids.reshape(-1)
    starts = (flat == 0).nonzero().reshape(-1).tolist()
    ends = [s for s in starts if s > 0] + [flat.numel()]
    return ends


def _module(name: str, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__long_vita_amd_standin__ = True
    return m


def install(force: bool = False):
    """Register the stand-in modules; returns the names installed.  With ``force`` an importable real module is
    shadowed too (tests)."""
    from . import ring_context_parallel as ring
    wanted = {
        "mindspeed": {},
        "mindspeed.utils": dict(get_actual_seq_len=state.get_actual_seq_len, set_actual_seq_len=state.set_actual_seq_len,
                                get_position_ids=state.get_position_ids, set_position_ids=state.set_position_ids,
                                compute_actual_seq_len=compute_actual_seq_len),
        "mindspeed.patch_utils": dict(MindSpeedPatchesManager=PatchesManager),
        "mindspeed.core": {},
        "mindspeed.core.context_parallel": {},
        "mindspeed.core.context_parallel.ring_context_parallel": dict(ringattn_context_parallel=ring.ringattn_context_parallel),
    }
    installed = []
    for name, attrs in wanted.items():
        if not force:
            try:
                importlib.import_module(name)
                continue
            except ImportError:
                pass
        mod = _module(name, **attrs)
        if name.count("."):
            parent, _, leaf = name.rpartition(".")
            if parent in sys.modules:
                setattr(sys.modules[parent], leaf, mod)
        if not attrs:
            mod.__path__ = []      # a package: submodules may be imported from it
        sys.modules[name] = mod
        installed.append(name)
    return installed

"""``@configclass``: dataclass wrapper used by every cfg object of the hot path.

Behavioural mirror of ``LAB/utils/configclass.py:31-113`` (own implementation):
* members without a type annotation become fields (type inferred from the default),
* mutable defaults (lists, dicts, other cfg instances) are deep-copied per instance instead of
  requiring ``field(default_factory=...)``,
* instances get ``to_dict / from_dict / replace / copy / validate``.
"""
from __future__ import annotations

import copy
import dataclasses
import inspect
import types
from dataclasses import MISSING, Field, dataclass, field
from typing import Any, Callable, ClassVar

_IMMUTABLE = (int, float, str, bool, bytes, type(None), tuple, frozenset, type, types.FunctionType, slice)


def _is_cfg(obj) -> bool:
    return dataclasses.is_dataclass(obj) and not isinstance(obj, type)


def _collect_annotations(cls):
    """Give every public, non-callable class attribute an annotation (so dataclass sees it)."""
    ann = dict(cls.__dict__.get("__annotations__", {}))
    # inherited annotations first, to keep field order base -> derived
    merged = {}
    for base in reversed(cls.__mro__[1:]):
        merged.update(getattr(base, "__annotations__", {}) if base is not object else {})
    for name, value in list(cls.__dict__.items()):
        if name.startswith("__") or name in ann:
            continue
        if name in merged and not isinstance(value, (classmethod, staticmethod, property)):
            ann[name] = merged[name]  # new default for an inherited field (e.g. ``func = uniform_noise``)
            continue
        if isinstance(value, (classmethod, staticmethod, property)) or inspect.isfunction(value):
            continue
        if isinstance(value, type) and name[:1].isupper():
            continue  # nested cfg *class* definitions (e.g. class PolicyCfg) are not fields
        ann[name] = type(value) if value is not MISSING else Any
    cls.__annotations__ = ann


def _wrap_mutable_defaults(cls):
    for name, ann in list(cls.__annotations__.items()):
        if ann is ClassVar or getattr(ann, "__origin__", None) is ClassVar or (isinstance(ann, str) and "ClassVar" in ann):
            continue
        if name not in cls.__dict__:
            # inherited field: dataclass re-uses the base Field
            continue
        default = cls.__dict__[name]
        if isinstance(default, Field):
            if default.default_factory is not MISSING and not callable(default.default_factory):
                value = default.default_factory
                default.default_factory = (lambda v: (lambda: copy.deepcopy(v)))(value)
            continue
        if default is MISSING:
            # a field whose value is the MISSING sentinel until the user sets it (checked by validate())
            setattr(cls, name, field(default_factory=lambda: MISSING))
            continue
        if isinstance(default, _IMMUTABLE) or callable(default) and not _is_cfg(default):
            continue
        setattr(cls, name, field(default_factory=(lambda v: (lambda: copy.deepcopy(v)))(default)))


def _post_init_copy(self):
    """Deep-copy mutable members that came from class-level defaults of parent classes."""
    for f in dataclasses.fields(self):
        value = getattr(self, f.name, MISSING)
        if value is MISSING or isinstance(value, _IMMUTABLE) or callable(value) and not _is_cfg(value):
            continue
        # values shared with the class default would alias between instances
        if f.default is not MISSING and value is f.default:
            setattr(self, f.name, copy.deepcopy(value))


def class_to_dict(obj) -> Any:
    if _is_cfg(obj):
        return {f.name: class_to_dict(getattr(obj, f.name)) for f in dataclasses.fields(obj)}
    if isinstance(obj, dict):
        return {k: class_to_dict(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(class_to_dict(v) for v in obj)
    if callable(obj) and not isinstance(obj, type) and hasattr(obj, "__module__") and hasattr(obj, "__name__"):
        return f"{obj.__module__}:{obj.__name__}"
    return obj


def update_class_from_dict(obj, data: dict, _ns: str = "") -> None:
    if not isinstance(data, dict):
        raise TypeError(f"Expected a dictionary to update '{_ns or type(obj).__name__}', received {type(data)}.")
    for key, value in data.items():
        full = f"{_ns}/{key}" if _ns else key
        is_map = isinstance(obj, dict)
        if not (key in obj if is_map else hasattr(obj, key)):
            raise KeyError(f"[Config]: Key not found under namespace: {full}.")
        current = obj[key] if is_map else getattr(obj, key)
        if isinstance(value, dict) and (_is_cfg(current) or isinstance(current, dict)):
            update_class_from_dict(current, value, full)
            continue
        if isinstance(current, (list, tuple)) and isinstance(value, (list, tuple)):
            if len(current) != len(value) and current is not None and len(current) > 0:
                raise ValueError(f"[Config]: Incorrect length under namespace: {full}. Expected: {len(current)}, Received: {len(value)}.")
            value = type(current)(value)
        elif callable(current) and isinstance(value, str):
            from .string import string_to_callable

            value = string_to_callable(value)
        elif current is not None and value is not None and not isinstance(value, type(current)) \
                and not (isinstance(current, float) and isinstance(value, int)):
            raise ValueError(
                f"[Config]: Incorrect type under namespace: {full}. Expected: {type(current)}, Received: {type(value)}."
            )
        if is_map:
            obj[key] = value
        else:
            setattr(obj, key, value)


def _validate(obj, prefix: str = "") -> list[str]:
    missing = []
    if _is_cfg(obj):
        items = ((f.name, getattr(obj, f.name, MISSING)) for f in dataclasses.fields(obj))
    elif isinstance(obj, dict):
        items = obj.items()
    elif isinstance(obj, (list, tuple)):
        items = ((str(i), v) for i, v in enumerate(obj))
    else:
        return missing
    for name, value in items:
        path = f"{prefix}.{name}" if prefix else str(name)
        if value is MISSING or (isinstance(value, type(MISSING))):
            missing.append(path)
        else:
            missing.extend(_validate(value, path))
    if prefix == "" and missing:
        raise TypeError(
            f"Missing values detected in object {type(obj).__name__} for the following fields:\n"
            + "\n".join(f"  - {m}" for m in missing)
        )
    return missing


def configclass(cls=None, **kwargs):
    def wrap(c):
        _collect_annotations(c)
        _wrap_mutable_defaults(c)
        user_post = c.__dict__.get("__post_init__")
        if user_post is not None:
            def __post_init__(self, *a, _u=user_post, **k):
                _post_init_copy(self)
                _u(self, *a, **k)
        else:
            inherited = getattr(c, "__post_init__", None)

            def __post_init__(self, *a, _i=inherited, **k):
                if _i is not None:
                    _i(self, *a, **k)
                else:
                    _post_init_copy(self)
        c.__post_init__ = __post_init__
        c.to_dict = lambda self: class_to_dict(self)
        c.from_dict = lambda self, data: update_class_from_dict(self, data, "")
        c.replace = lambda self, **kw: dataclasses.replace(self, **kw)
        c.copy = lambda self: copy.deepcopy(self)
        c.validate = lambda self, prefix="": _validate(self, prefix)
        kwargs.setdefault("eq", True)
        return dataclass(c, **kwargs)

    return wrap if cls is None else wrap(cls)

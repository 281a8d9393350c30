"""Name resolution helpers (host logic, no tensors).

Mirror of ``LAB/utils/string.py:138-368``: ``"module:attr"`` strings to callables and regular
expression -> index resolution used by actuator cfgs and ``SceneEntityCfg``.
"""
from __future__ import annotations

import ast
import importlib
import re
from collections.abc import Callable, Sequence
from typing import Any


def is_lambda_expression(name: str) -> bool:
    try:
        body = ast.parse(name).body
    except SyntaxError:
        return False
    return len(body) == 1 and isinstance(body[0], ast.Expr) and isinstance(body[0].value, ast.Lambda)


def callable_to_string(value: Callable) -> str:
    if not callable(value):
        raise ValueError(f"The input argument is not callable: {value}.")
    return f"{value.__module__}:{value.__name__}"


def string_to_callable(name: str) -> Callable:
    try:
        if is_lambda_expression(name):
            obj = eval(name)  # noqa: S307 - same contract as the reference (lambda strings in cfg files)
        else:
            mod_name, attr_name = name.split(":")
            obj = getattr(importlib.import_module(mod_name), attr_name)
        if callable(obj):
            return obj
        raise AttributeError(f"The imported object is not callable: '{name}'")
    except (ValueError, ModuleNotFoundError) as e:
        raise ValueError(
            f"Could not resolve the input string '{name}' into callable object."
            f" The format of input should be 'module:attribute_name'.\nReceived the error:\n {e}."
        )


def _match(keys: Sequence[str], values: Sequence[Any] | None, targets: Sequence[str], preserve_order: bool):
    owner: list[str | None] = [None] * len(targets)
    hits: list[tuple[int, int]] = []  # (key index, target index) in target-major discovery order
    found = [[] for _ in keys]
    for ti, target in enumerate(targets):
        for ki, key in enumerate(keys):
            if re.fullmatch(key, target):
                if owner[ti] is not None:
                    raise ValueError(f"Multiple matches for '{target}': '{owner[ti]}' and '{key}'!")
                owner[ti] = key
                hits.append((ki, ti))
                found[ki].append(target)
    if preserve_order:
        hits.sort(key=lambda kt: kt[0])  # stable: key order first, target order within a key
    if not all(found):
        listing = "".join(f"\t{k}: {f}\n" for k, f in zip(keys, found))
        raise ValueError(
            "Not all regular expressions are matched! Please check that the regular expressions are correct: \n"
            f"{listing}Available strings: {list(targets)}\n"
        )
    idx = [ti for _, ti in hits]
    names = [targets[ti] for ti in idx]
    vals = [values[ki] for ki, _ in hits] if values is not None else None
    return idx, names, vals


def resolve_matching_names(keys: str | Sequence[str], list_of_strings: Sequence[str], preserve_order: bool = False):
    if isinstance(keys, str):
        keys = [keys]
    idx, names, _ = _match(list(keys), None, list_of_strings, preserve_order)
    return idx, names


def resolve_matching_names_values(data: dict[str, Any], list_of_strings: Sequence[str], preserve_order: bool = False):
    if not isinstance(data, dict):
        raise TypeError(f"Input argument `data` should be a dictionary. Received: {data}")
    idx, names, vals = _match(list(data.keys()), list(data.values()), list_of_strings, preserve_order)
    return idx, names, vals

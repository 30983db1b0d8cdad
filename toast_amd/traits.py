"""A small trait system with the surface of the reference's ``TraitConfig``
(src/toast/traits.py:244-338) -- traitlets itself is not a dependency here.

Every Operator / Template is a ``TraitConfig``: class-level ``Trait`` declarations, keyword
construction, ``name`` / ``enabled`` / ``kernel_implementation`` common traits,
``_validate_<trait>`` and ``_observe_<trait>`` hooks, and ``select_kernels``.
"""

import enum


class ImplementationType(enum.IntEnum):
    """Kernel implementations (reference: src/toast/accelerator/kernel_registry.py:12-24)."""

    DEFAULT = 0
    COMPILED = 1
    NUMPY = 2
    JAX = 3


class TraitError(Exception):
    pass


_NODEFAULT = object()


class Trait:
    """Descriptor holding a typed, documented, validated attribute."""

    def __init__(self, default=None, help="", allow_none=False, kind=None, klass=None):
        self.default = default
        self.help = help
        self.allow_none = allow_none or default is None
        self.kind = kind
        self.klass = klass
        self.name = None

    def __set_name__(self, owner, name):
        self.name = name

    def __get__(self, obj, objtype=None):
        if obj is None:
            return self
        if self.name not in obj.__dict__.setdefault("_trait_values", {}):
            default = self.default
            if isinstance(default, (list, dict, set)):
                default = type(default)(default)
            obj._trait_values[self.name] = default
        return obj._trait_values[self.name]

    def _coerce(self, value):
        if value is None:
            if not self.allow_none:
                raise TraitError(f"The '{self.name}' trait does not allow None")
            return None
        if self.kind is bool:
            if not isinstance(value, (bool,)) and not (hasattr(value, "dtype") and value.dtype == bool):
                raise TraitError(f"The '{self.name}' trait expected a bool, not {value!r}")
            return bool(value)
        if self.kind is int:
            if isinstance(value, bool) or not isinstance(value, (int,)) and not hasattr(value, "__index__"):
                raise TraitError(f"The '{self.name}' trait expected an int, not {value!r}")
            return int(value)
        if self.kind is float:
            return float(value)
        if self.kind is str:
            if not isinstance(value, str):
                raise TraitError(f"The '{self.name}' trait expected a unicode string, not {value!r}")
            return value
        if self.kind is list:
            return list(value)
        if self.klass is not None and not isinstance(value, self.klass):
            raise TraitError(f"The '{self.name}' trait expected a {self.klass.__name__} instance")
        return value

    def __set__(self, obj, value):
        value = self._coerce(value)
        validator = getattr(obj, "_validate_" + self.name, None)
        if validator is not None:
            value = validator(value)
        values = obj.__dict__.setdefault("_trait_values", {})
        old = values.get(self.name, self.default)
        values[self.name] = value
        observer = getattr(obj, "_observe_" + self.name, None)
        if observer is not None:
            observer({"name": self.name, "old": old, "new": value})


def Int(default=0, help="", allow_none=False):
    return Trait(default, help, allow_none, kind=int)


def Bool(default=False, help="", allow_none=False):
    return Trait(default, help, allow_none, kind=bool)


def Float(default=0.0, help="", allow_none=False):
    return Trait(default, help, allow_none, kind=float)


def Unicode(default=None, help="", allow_none=False):
    return Trait(default, help, allow_none, kind=str)


def List(default=None, help="", allow_none=False):
    return Trait([] if default is None else default, help, allow_none, kind=list)


def Instance(klass=None, default=None, help="", allow_none=True):
    return Trait(default, help, allow_none, klass=klass)


def Any(default=None, help="", allow_none=True):
    return Trait(default, help, allow_none)


class TraitConfig:
    """Base class of configurable objects (reference: src/toast/traits.py:244-338)."""

    name = Unicode(None, allow_none=True, help="The 'name' of this class instance")
    enabled = Bool(True, help="If True, this class instance is marked as enabled")
    kernel_implementation = Any(ImplementationType.DEFAULT, help="Which kernel implementation to use")

    def __init__(self, **kwargs):
        self._trait_values = {}
        if kwargs.get("name") is None:
            kwargs["name"] = type(self).__name__
        for key, val in kwargs.items():
            if not self.has_trait(key):
                raise TraitError(f"Class {type(self).__name__} has no trait '{key}'")
        # apply in declaration order so that validators see earlier traits
        for key in self.trait_names():
            if key in kwargs:
                setattr(self, key, kwargs[key])

    @classmethod
    def trait_names(cls):
        names = []
        for klass in reversed(cls.__mro__):
            for key, val in vars(klass).items():
                if isinstance(val, Trait) and key not in names:
                    names.append(key)
        return names

    def has_trait(self, name):
        return isinstance(getattr(type(self), name, None), Trait)

    def traits(self):
        return {k: getattr(self, k) for k in self.trait_names()}

    def __repr__(self):
        body = ", ".join(f"{k}={v!r}" for k, v in self.traits().items())
        return f"<{type(self).__name__} {body}>"

    # --- kernel selection (src/toast/traits.py:286-338)
    def _implementations(self):
        return [ImplementationType.DEFAULT]

    def implementations(self):
        return self._implementations()

    def _supports_accel(self):
        return False

    def supports_accel(self):
        return self._supports_accel()

    def select_kernels(self, use_accel=None):
        """Return ``(implementation, use_accel)`` for a call.

        ``use_accel=None`` or ``False`` -> ``(DEFAULT, False)``: buffers are host buffers (they
        are staged through the GPU by the library; there is no CPU kernel).  ``True`` ->
        ``(COMPILED, True)`` and the operator must support accelerators.
        """
        if use_accel:
            if not self.supports_accel():
                raise RuntimeError(f"Operator {self.name} does not support accelerators")
            impls = self.implementations()
            if ImplementationType.COMPILED not in impls and ImplementationType.DEFAULT not in impls:
                raise RuntimeError(f"Operator {self.name} has no compiled accelerator kernels")
            return ImplementationType.COMPILED, True
        return ImplementationType.DEFAULT, False

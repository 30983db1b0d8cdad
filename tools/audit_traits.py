"""Compare the traits (names and default values) of every operator / template class of toast_amd with the class of
the same name in the reference sources (parsed with ast: nothing of the reference is imported).  Container only."""
import ast
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/src/toast"
TRAITS = {"Int", "Float", "Bool", "Unicode", "Instance", "List", "Quantity", "Unit", "Tuple", "Set", "Dict", "Callable",
          "UseEnum", "Enum"}


def ref_classes():
    out = {}
    for path in glob.glob(REF + "/ops/**/*.py", recursive=True) + glob.glob(REF + "/templates/*.py"):
        try:
            tree = ast.parse(open(path).read())
        except SyntaxError:
            continue
        for node in ast.walk(tree):
            if isinstance(node, ast.ClassDef):
                traits = {}
                for st in node.body:
                    if isinstance(st, ast.Assign) and len(st.targets) == 1 and isinstance(st.targets[0], ast.Name) \
                            and isinstance(st.value, ast.Call):
                        fn = st.value.func
                        name = fn.id if isinstance(fn, ast.Name) else getattr(fn, "attr", None)
                        if name in TRAITS:
                            default = ast.unparse(st.value.args[0]) if st.value.args else None
                            for kw in st.value.keywords:
                                if kw.arg == "default_value":
                                    default = ast.unparse(kw.value)
                            traits[st.targets[0].id] = (name, default)
                if traits:
                    out.setdefault(node.name, (path, traits))
    return out


def ours():
    import toast_amd.ops as ops
    import toast_amd.templates as templates
    from toast_amd.traits import TraitConfig

    found = {}
    for mod in (ops, templates):
        for name in dir(mod):
            obj = getattr(mod, name)
            if isinstance(obj, type) and issubclass(obj, TraitConfig):
                found[name] = obj
    return found


def norm(v):
    if v is None:
        return "None"
    v = str(v).replace("u.", "").replace(" ", "")
    for a, b in (("'", '"'),):
        v = v.replace(a, b)
    return v


def differences():
    """[(class, "MISSING" | "DEFAULT", trait, reference default, our default)] for every class present on both sides."""
    from toast_amd.data import defaults
    from toast_amd.traits import Trait

    ref = ref_classes()
    mine = ours()
    out = []
    for cname, cls in sorted(mine.items()):
        if cname not in ref:
            continue
        _, rtraits = ref[cname]
        inst_traits = {}
        for klass in reversed(cls.__mro__):
            for k, v in vars(klass).items():
                if isinstance(v, Trait):
                    inst_traits[k] = v
        for tname, (ttype, rdef) in sorted(rtraits.items()):
            if tname not in inst_traits:
                out.append((cname, "MISSING", tname, rdef, None))
                continue
            odef = inst_traits[tname].default
            # evaluate the reference default where it is a plain literal or a defaults.* name
            try:
                rval = eval(rdef, {"defaults": defaults, "None": None, "True": True, "False": False}) if rdef else None
                known = True
            except Exception:
                rval, known = rdef, False
            if known:
                same = (rval == odef) or (rval is None and odef is None)
                if not same and isinstance(rval, (int, float)) and isinstance(odef, (int, float)):
                    same = float(rval) == float(odef)
            else:
                same = norm(rdef) in norm(repr(odef)) or norm(repr(odef)) in norm(rdef)
            if not same:
                out.append((cname, "DEFAULT", tname, rdef, odef))
    return out


def main():
    diffs = differences()
    for cname, kind, tname, rdef, odef in diffs:
        print(f"{cname:28s} {kind:8s} {tname}: reference {rdef!r}" + ("" if kind == "MISSING" else f"  ours {odef!r}"))
    print("differences:", len(diffs))


if __name__ == "__main__":
    main()

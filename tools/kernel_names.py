"""Kernel names of the library as the tools print them."""
import re


def demangle(n):
    """y2:: kernel names without a demangler that knows _Float16 (DF16_): _ZN2y2<len><name>[I<args>E]... -> name<args>"""
    m = re.match(r"_ZN2y2(\d+)", n)
    if not m:
        return n.split("(")[0].replace("void ", "")
    ln = int(m.group(1)); name = n[m.end():m.end() + ln]; rest = n[m.end() + ln:]
    if rest.startswith("I"):
        args = re.findall(r"L([ib])(n?\d+)E", rest[:rest.index("EE") + 1] if "EE" in rest else rest)
        name += "<" + ",".join(("true" if v == "1" else "false") if t == "b" else v.replace("n", "-") for t, v in args) + ">"
    return "y2::" + name

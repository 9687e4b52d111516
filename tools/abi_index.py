#!/usr/bin/env python3
"""Index of the C ABI: every entry point include/yolo2_hip.h declares, with the header line, the tier it is declared under and the
first sentence of the comment in front of it (which cites the reference interface it stands for, where there is one).  Writes the table
between the two markers of INTEGRATION.md's last section; tests/test_host_logic.py checks that the table is current.
usage: python3 tools/abi_index.py [--check]"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "yolo2_hip.h")
DOC = os.path.join(ROOT, "INTEGRATION.md")
BEGIN, END = "<!-- abi-index:begin (tools/abi_index.py) -->", "<!-- abi-index:end -->"
PROTO = re.compile(r"^\s*(?:const\s+)?[A-Za-z_][A-Za-z0-9_]*[\s\*]+\**((?:yolo2|memory|dma_buffer)_[a-z0-9_]+)\s*\(")
REF = re.compile(r"\(?((?:linux_app|hls|src|include)/[A-Za-z0-9_./]+(?::[0-9,\- ]+)?|[a-z0-9_]+\.(?:h|hpp|c|cpp):[0-9\-,]+)")


def first_sentence(text):
    text = re.sub(r"\s+", " ", text).strip()
    m = re.search(r"(.+?[.:])(?: |$)(?=[A-Z(]|$)", text)
    s = (m.group(1) if m else text).rstrip(":")
    return s if len(s) <= 230 else s[:227] + "..."


def entries():
    lines = open(HDR).read().split("\n")
    tier, comment, in_comment, out, depth = "", [], False, [], 0
    last_comment = ""
    for no, ln in enumerate(lines, 1):
        s = ln.strip()
        if s.startswith("/* ----") or s.startswith("/* ===="):
            tier = re.sub(r"[-=/\*]+", " ", s).strip().rstrip(".")
            last_comment = ""
            if "*/" not in s:            # the section's comment goes on: its text describes the entries that follow
                in_comment, comment = True, []
            continue
        if in_comment:
            comment.append(s.lstrip("* ").rstrip("*/ "))
            if "*/" in s:
                in_comment = False
                last_comment = " ".join(comment)
            continue
        if s.startswith("/*"):
            comment = [s[2:].replace("*/", "").strip()]
            if "*/" in s:
                last_comment = comment[0]
            else:
                in_comment = True
            continue
        if depth == 0:
            m = PROTO.match(ln)
            if m and not s.startswith(("typedef", "#", "return")):
                trailing = re.search(r"/\*\s*(.*?)\s*\*/", ln)
                text = last_comment or (trailing.group(1) if trailing else "")
                if trailing and not last_comment:
                    text = trailing.group(1)
                out.append((m.group(1), no, tier, first_sentence(text) or "(no comment of its own: see the section)"))
        depth += s.count("{") - s.count("}") if not s.startswith("extern") else 0
        depth = max(depth, 0)
    return out


def table():
    rows = ["| entry point | `include/yolo2_hip.h` | declared under | what the header says (first sentence) |", "|---|---|---|---|"]
    for name, no, tier, text in entries():
        rows.append(f"| `{name}` | :{no} | {tier} | {text.replace('|', '/')} |")
    return "\n".join(rows)


def main():
    doc = open(DOC).read()
    if BEGIN not in doc or END not in doc:
        sys.exit(f"{DOC}: markers not found")
    new = doc[:doc.index(BEGIN) + len(BEGIN)] + "\n" + table() + "\n" + doc[doc.index(END):]
    if "--check" in sys.argv:
        sys.exit(0 if new == doc else "INTEGRATION.md's ABI index is stale: run python3 tools/abi_index.py")
    open(DOC, "w").write(new)
    print(f"{len(entries())} entry points indexed")


if __name__ == "__main__":
    main()

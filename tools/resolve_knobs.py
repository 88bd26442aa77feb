"""Resolve retired compile-time knobs in a C++ source: a small `unifdef`.

    python tools/resolve_knobs.py FILE NAME=VALUE [NAME=VALUE ...]     (rewrites FILE in place)

Every `#if` / `#ifdef` / `#ifndef` / `#elif` whose condition mentions ONLY the given names is evaluated and the dead branch is
deleted (the directive lines go too); conditions that mention any other name are left alone.  `#ifndef NAME / #define NAME v /
#endif` default blocks of a given name are deleted as well.  Uses of the names in ordinary code are NOT touched: replace them by
hand (a constexpr or the literal) before or after.  Used in round 5 to move the measured losers of qp_resident.hpp out of the
product kernel (their patches live in tools/experiments/)."""
import re
import sys


def evaluate(expr, vals):
    names = set(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", expr)) - {"defined"}
    if not names or not names <= set(vals):
        return None
    e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", lambda m: "1", expr)
    e = re.sub(r"[A-Za-z_][A-Za-z0-9_]*", lambda m: str(vals[m.group(0)]), e)
    e = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", "!=")
    return bool(eval(e))          # noqa: S307 -- integer expressions of the knobs only


def main():
    path = sys.argv[1]
    vals = {}
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        vals[k] = int(v)
    lines = open(path).read().split("\n")
    out = []
    # stack entries: [kind, emitting, taken, resolved] ; kind 'r' resolved conditional, 'k' kept conditional
    stack = []
    i = 0

    def emitting():
        return all(s[1] for s in stack)

    while i < len(lines):
        ln = lines[i]
        st = ln.strip()
        m = re.match(r"#\s*(ifndef|ifdef|if|elif|else|endif)\b(.*)", st)
        if not m:
            if emitting():
                out.append(ln)
            i += 1
            continue
        d, rest = m.group(1), m.group(2).split("//")[0].strip()
        if d in ("if", "ifdef", "ifndef"):
            if d == "if":
                val = evaluate(rest, vals)
            else:
                name = rest.split()[0]
                val = None
                if name in vals:
                    # `#ifndef NAME` default block: drop it entirely (the name is defined by us)
                    val = (d == "ifdef")
            if val is None:
                stack.append(["k", True, False])
                if emitting():
                    out.append(ln)
            else:
                stack.append(["r", val, val])
        elif d == "elif":
            top = stack[-1]
            if top[0] == "k":
                if emitting():
                    out.append(ln)
            else:
                val = evaluate(rest, vals)
                if val is None:
                    raise SystemExit(f"{path}:{i + 1}: #elif mixes resolved and unresolved names")
                top[1] = (not top[2]) and val
                top[2] = top[2] or val
        elif d == "else":
            top = stack[-1]
            if top[0] == "k":
                if emitting():
                    out.append(ln)
            else:
                top[1] = not top[2]
                top[2] = True
        else:
            top = stack.pop()
            if top[0] == "k" and emitting():
                out.append(ln)
        i += 1
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main()

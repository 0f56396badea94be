"""Extracts the `statement error` cases of the reference's *_negative.test files (the table functions of this path)
into tests/golden/negative_cases.json: function, positional and named arguments as JSON values, the substring the
error message must contain, and the file:line the case stands at.  Run in the build container (reads
/root/reference/test/sql as text); the fixture it writes is what the tests load.

    python tests/golden/make_negative_cases.py
"""
import glob
import json
import os
import re

REF = "/root/reference/test/sql"
HERE = os.path.dirname(os.path.abspath(__file__))
FUNCTIONS = ("plink_freq", "plink_hardy", "plink_missing", "plink_score", "plink_pca", "plink_ld", "read_pgen", "read_pfile")


class Unparsed(Exception):
    pass


def parse_value(s, i):
    """One SQL literal starting at s[i] -> (python value, next index)."""
    while s[i].isspace():
        i += 1
    c = s[i]
    if c == "'":
        j, out = i + 1, []
        while True:
            if s[j] == "'" and s[j + 1:j + 2] == "'":
                out.append("'"); j += 2
            elif s[j] == "'":
                break
            else:
                out.append(s[j]); j += 1
        v, i = "".join(out), j + 1
    elif c == "[":
        v, i = [], i + 1
        while True:
            while s[i].isspace():
                i += 1
            if s[i] == "]":
                i += 1
                break
            item, i = parse_value(s, i)
            v.append(item)
            while s[i].isspace():
                i += 1
            if s[i] == ",":
                i += 1
    elif c == "{":
        v, i = {}, i + 1
        while True:
            while s[i].isspace():
                i += 1
            if s[i] == "}":
                i += 1
                break
            bare = re.match(r"[A-Za-z_][A-Za-z_0-9]*", s[i:])  # {start: 0}: DuckDB takes unquoted field names
            if bare:
                k, i = bare.group(0), i + len(bare.group(0))
            else:
                k, i = parse_value(s, i)
            while s[i].isspace():
                i += 1
            if s[i] != ":":
                raise Unparsed(s[i:i + 20])
            val, i = parse_value(s, i + 1)
            v[k] = val
            while s[i].isspace():
                i += 1
            if s[i] == ",":
                i += 1
    else:
        m = re.match(r"(?i)(true|false|null|-?\d+\.\d*(?:e-?\d+)?|-?\.\d+|-?\d+(?:e-?\d+)?)", s[i:])
        if not m:
            raise Unparsed(s[i:i + 30])
        t = m.group(1).lower()
        v = True if t == "true" else False if t == "false" else None if t == "null" else (
            float(t) if any(ch in t for ch in ".e") else int(t))
        i += len(m.group(1))
    m = re.match(r"\s*::\s*[A-Za-z_]+(?:\s*\[\s*\]|\s*\([^)]*\))*", s[i:])  # a cast keeps the JSON value
    if m:
        i += len(m.group(0))
    return v, i


def parse_call(sql):
    m = re.search(r"(?is)\bFROM\s+(" + "|".join(FUNCTIONS) + r")\s*\(", sql)
    if not m or re.search(r"(?is)\bFROM\b", sql[m.end():]) or re.search(r"(?is)\b(JOIN|WHERE)\b", sql):
        raise Unparsed("not a single call")
    head = sql[:m.start()].strip()
    fn, i, args, named = m.group(1), m.end(), [], {}
    while True:
        while sql[i].isspace():
            i += 1
        if sql[i] == ")":
            break
        k = re.match(r"([A-Za-z_][A-Za-z_0-9]*)\s*:=", sql[i:])
        if k:
            v, i = parse_value(sql, i + len(k.group(0)))
            named[k.group(1)] = v
        else:
            v, i = parse_value(sql, i)
            args.append(v)
        while sql[i].isspace():
            i += 1
        if sql[i] == ",":
            i += 1
    if sql[i + 1:].strip().rstrip(";").strip():
        raise Unparsed("text behind the call")
    cols = None
    sel = re.match(r"(?is)SELECT\s+(.*)$", head)
    if sel and sel.group(1).strip() not in ("*", "COUNT(*)", "count(*)"):
        names = [c.strip() for c in sel.group(1).split(",")]
        if all(re.fullmatch(r"[A-Za-z_][A-Za-z_0-9]*", c) for c in names):
            cols = names
        else:
            raise Unparsed("select list")
    return fn, args, named, cols


def main():
    cases, skipped = [], []
    for path in sorted(glob.glob(os.path.join(REF, "*_negative.test"))):
        name = os.path.basename(path)
        if not name.startswith(FUNCTIONS):
            continue
        lines = open(path).read().splitlines()
        settings, i = {}, 0
        while i < len(lines):
            if lines[i].strip() == "statement ok":
                stmt = lines[i + 1].strip()
                m = re.match(r"(?i)SET\s+(\w+)\s*=\s*(.+?);?$", stmt)
                r = re.match(r"(?i)RESET\s+(\w+)", stmt)
                if m:
                    settings[m.group(1)] = parse_value(m.group(2), 0)[0]
                elif r:
                    settings.pop(r.group(1), None)
                i += 2
                continue
            if lines[i].strip() != "statement error":
                i += 1
                continue
            at = i + 1
            j = i + 1
            while lines[j].strip() != "----":
                j += 1
            sql = "\n".join(lines[i + 1:j])
            k = j + 1
            want = []
            while k < len(lines) and lines[k].strip():
                want.append(lines[k])
                k += 1
            i = k
            try:
                fn, args, named, cols = parse_call(sql)
            except (Unparsed, IndexError) as e:
                skipped.append(f"{name}:{at}: {e}")
                continue
            case = {"source": f"test/sql/{name}:{at}", "function": fn, "args": args, "named": named,
                    "error_contains": " ".join(w.strip() for w in want)}
            if cols:
                case["columns"] = cols
            if settings:
                case["settings"] = dict(settings)
            cases.append(case)
    with open(os.path.join(HERE, "negative_cases.json"), "w") as f:
        json.dump({"_from": "tests/golden/make_negative_cases.py over /root/reference/test/sql/*_negative.test",
                   "cases": cases, "not_extracted": skipped}, f, indent=1)
        f.write("\n")
    print(len(cases), "cases;", len(skipped), "not extracted")
    for s in skipped:
        print("  ", s)


if __name__ == "__main__":
    main()

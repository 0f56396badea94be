"""Extracts the `query` blocks of the reference's test/sql/*.test files that are ONE call of one of this path's table
functions with a select list, WHERE, ORDER BY and LIMIT simple enough for tests/sqlmini.py to evaluate, into
tests/golden/query_cases.json: the call (function, arguments), the clauses as text, the column-type string and the
expected rows exactly as the reference's test states them.  Run in the build container (reads /root/reference/test/sql
as text); the fixture is what the tests load.

    python tests/golden/make_query_cases.py
"""
import glob
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import sqlmini  # noqa: E402
from make_negative_cases import FUNCTIONS, Unparsed, parse_value  # noqa: E402

REF = "/root/reference/test/sql"
# files about things outside this path (other readers, remote / parquet / glob file access, regression models)
SKIP_FILES = re.compile(r"^(plink_glm|read_plink_vcf|read_pvar|read_psam|parquet_|read_pfile_psam_parquet|"
                        r"read_pfile_region_parquet|read_pfile_vfs|read_pgen_vfs|read_pfile_glob|read_pfile_localize|"
                        r"read_file_search_path|flexible_companions)")


def split_call(sql):
    m = re.search(r"(?is)\bFROM\s+(" + "|".join(FUNCTIONS) + r")\s*\(", sql)
    if not m:
        raise Unparsed("no call of this path")
    if len(re.findall(r"(?i)\bFROM\b", sql)) != 1 or re.search(r"(?i)\b(JOIN|GROUP\s+BY|UNION|EXCEPT|WITH|HAVING|OVER|DISTINCT|UNNEST)\b", sql):
        raise Unparsed("more than one relation / grouping")
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
    head = re.match(r"(?is)\s*SELECT\s+(.*)$", sql[:m.start()])
    if not head:
        raise Unparsed("no select list")
    tail = sql[i + 1:].strip().rstrip(";").strip()
    tail = re.sub(r"(?is)^(AS\s+)?[a-z_][a-z_0-9]*\s*(?=(WHERE|ORDER|LIMIT|$))", "", tail) if not re.match(r"(?i)(WHERE|ORDER|LIMIT)\b", tail) else tail
    mm = re.match(r"(?is)^(?:WHERE\s+(?P<where>.*?))?\s*(?:ORDER\s+BY\s+(?P<order>.*?))?\s*(?:LIMIT\s+(?P<limit>\d+))?\s*$", tail)
    if not mm:
        raise Unparsed("tail: " + tail[:40])
    return fn, args, named, head.group(1).strip(), mm.group("where"), mm.group("order"), mm.group("limit")


def main():
    cases, skipped = [], []
    for path in sorted(glob.glob(os.path.join(REF, "*.test"))):
        name = os.path.basename(path)
        if SKIP_FILES.match(name):
            continue
        lines = open(path).read().splitlines()
        settings, i = {}, 0
        while i < len(lines):
            head = lines[i].strip()
            if head == "statement ok":
                stmt = lines[i + 1].strip()
                m = re.match(r"(?i)SET\s+(\w+)\s*=\s*(.+?);?$", stmt)
                r = re.match(r"(?i)RESET\s+(\w+)", stmt)
                try:
                    if m:
                        settings[m.group(1)] = parse_value(m.group(2), 0)[0]
                    elif r:
                        settings.pop(r.group(1), None)
                except Unparsed:
                    pass
                i += 2
                continue
            if not head.startswith("query"):
                i += 1
                continue
            at = i + 1
            parts = head.split()
            types = parts[1] if len(parts) > 1 else ""
            rowsort = "rowsort" in parts[2:]
            j = i + 1
            while lines[j].strip() != "----":
                j += 1
            sql = " ".join(x.strip() for x in lines[i + 1:j] if not x.strip().startswith("--"))
            k = j + 1
            rows = []
            while k < len(lines) and lines[k] != "":
                rows.append(lines[k].split("\t"))
                k += 1
            i = k
            if not re.search(r"\b(" + "|".join(FUNCTIONS) + r")\s*\(", sql):
                continue
            try:
                fn, args, named, select, where, order, limit = split_call(sql)
                sqlmini.compile_query(select, where, order)  # raises sqlmini.Unsupported
            except (Unparsed, IndexError, sqlmini.Unsupported) as e:
                skipped.append(f"{name}:{at}: {e}")
                continue
            paths = [a for a in args if isinstance(a, str)] + [x for a in args if isinstance(a, list) for x in a if isinstance(x, str)]
            missing = [q for q in paths if q.startswith("test/data/") and not (
                os.path.exists(os.path.join(HERE, "data", q[10:])) or os.path.exists(os.path.join(HERE, "data", q[10:] + ".pgen")))]
            if missing:
                skipped.append(f"{name}:{at}: fixture not in the reference tree: {missing[0]}")
                continue
            case = {"source": f"test/sql/{name}:{at}", "function": fn, "args": args, "named": named, "select": select,
                    "types": types, "expected": rows}
            for key, val in (("where", where), ("order_by", order), ("limit", int(limit) if limit else None),
                             ("rowsort", rowsort or None), ("settings", dict(settings) or None)):
                if val:
                    case[key] = val
            cases.append(case)
    with open(os.path.join(HERE, "query_cases.json"), "w") as f:
        json.dump({"_from": "tests/golden/make_query_cases.py over /root/reference/test/sql/*.test",
                   "cases": cases, "not_extracted": skipped}, f, indent=0)
        f.write("\n")
    print(len(cases), "cases;", len(skipped), "not extracted")
    import collections
    why = collections.Counter(s.split(": ", 1)[1][:50] for s in skipped)
    for w, n in why.most_common(25):
        print(f"  {n:4d} {w}")


if __name__ == "__main__":
    main()

"""Extracts the `query` blocks of the reference's test/sql/*.test files whose relations are all calls of this path's
table functions and whose SQL tests/sqlmini.py can evaluate (select lists, WHERE, GROUP BY, ORDER BY, LIMIT, subqueries
in FROM and as scalars, DESCRIBE) into tests/golden/query_cases.json: the statement, the column-type string and the
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


def call_paths(calls):
    out = []
    for _, _, args, named in calls:
        for node in list(args) + [v for k, v in named.items() if k in ("pvar", "psam", "pgen", "sex_file")]:
            v = sqlmini.literal(node)
            out.extend(x for x in (v if isinstance(v, list) else [v]) if isinstance(x, str))
    return out


def main():
    cases, skipped = [], []
    for path in sorted(glob.glob(os.path.join(REF, "*.test"))):
        name = os.path.basename(path)
        if SKIP_FILES.match(name):
            continue
        lines = open(path).read().splitlines()
        settings, variables, i = {}, {}, 0
        while i < len(lines):
            head = lines[i].strip()
            if head == "statement ok":
                stmt = lines[i + 1].strip()
                var = re.match(r"(?i)SET\s+VARIABLE\s+(\w+)\s*=\s*(.+?);?$", stmt)
                if var:
                    try:
                        variables[var.group(1)] = parse_value(var.group(2), 0)[0]
                    except Unparsed:
                        pass
                    i += 2
                    continue
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
                q = sqlmini.parse_sql(sql)
                calls = sqlmini.check_select(q, FUNCTIONS + ("read_pvar", "read_psam"))
                if all(c[1] in ("read_pvar", "read_psam") for c in calls):
                    raise sqlmini.Unsupported("no call of this path")
                paths = call_paths(calls)
            except (Unparsed, IndexError, sqlmini.Unsupported, KeyError, TypeError) as e:
                skipped.append(f"{name}:{at}: {e}")
                continue
            missing = [q_ for q_ in paths if q_.startswith("test/data/") and not (
                os.path.exists(os.path.join(HERE, "data", q_[10:])) or os.path.exists(os.path.join(HERE, "data", q_[10:] + ".pgen")))]
            if missing:
                skipped.append(f"{name}:{at}: fixture not in the reference tree: {missing[0]}")
                continue
            case = {"source": f"test/sql/{name}:{at}", "sql": sql, "types": types, "expected": rows}
            for key, val in (("rowsort", rowsort or None), ("settings", dict(settings) or None),
                             ("variables", dict(variables) if "getvariable" in sql else None)):
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

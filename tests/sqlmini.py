"""A small evaluator for the SQL around a table-function call in the reference's .test files: select lists of columns,
struct fields, list elements, a few scalar functions and aggregates; WHERE with comparisons, IS NULL, IN, AND / OR;
ORDER BY; LIMIT -- over the rows the shells return (dicts of Python values).  Test infrastructure: it stands in for
DuckDB's executor so that the reference's own queries and expected rows (tests/golden/query_cases.json) can be run
against the GPU shells.  Anything outside its grammar raises Unsupported and the case is not extracted."""
import math
import re


class Unsupported(Exception):
    pass


TOKEN = re.compile(r"\s*(?:(?P<num>\d+\.\d*(?:[eE][-+]?\d+)?|\.\d+|\d+(?:[eE][-+]?\d+)?)|(?P<str>'(?:[^']|'')*')|"
                   r"(?P<id>[A-Za-z_][A-Za-z_0-9]*|\"[^\"]+\")|(?P<op><>|!=|<=|>=|->|::|:=|[-+*/%(),.\[\]<>={}:;]))")
AGGREGATES = {"count", "count_distinct", "sum", "min", "max", "avg", "bool_and", "bool_or", "list", "first", "any_value"}


def tokenize(text):
    out, i = [], 0
    text = text.strip()
    while i < len(text):
        m = TOKEN.match(text, i)
        if not m or m.end() == i:
            raise Unsupported("token at " + text[i:i + 20])
        kind = m.lastgroup
        out.append((kind, m.group(kind)))
        i = m.end()
    return out


class Parser:
    def __init__(self, text):
        self.t = tokenize(text)
        self.i = 0

    def peek(self, k=0):
        return self.t[self.i + k] if self.i + k < len(self.t) else (None, None)

    def kw(self, *words):
        kind, v = self.peek()
        return kind == "id" and v.lower() in words

    def take(self, kind=None, val=None):
        k, v = self.peek()
        if (kind and k != kind) or (val is not None and (v or "").lower() != val):
            raise Unsupported(f"expected {val or kind} at {v}")
        self.i += 1
        return v

    def done(self):
        return self.i >= len(self.t)

    # expression grammar: or > and > not > comparison > additive > multiplicative > unary > postfix > primary
    def expr(self):
        left = self.and_()
        while self.kw("or"):
            self.take()
            left = ("or", left, self.and_())
        return left

    def and_(self):
        left = self.not_()
        while self.kw("and"):
            self.take()
            left = ("and", left, self.not_())
        return left

    def not_(self):
        if self.kw("not"):
            self.take()
            return ("not", self.not_())
        return self.cmp()

    def cmp(self):
        left = self.add()
        k, v = self.peek()
        if k == "op" and v in ("=", "<>", "!=", "<", "<=", ">", ">="):
            self.take()
            return ("cmp", v, left, self.add())
        if self.kw("is"):
            self.take()
            neg = False
            if self.kw("not"):
                self.take()
                neg = True
            if self.kw("null"):
                self.take()
                return ("isnull", neg, left)
            if self.kw("distinct"):
                self.take()
                self.take("id", "from")
                return ("distinct", not neg, left, self.add())
            raise Unsupported("IS ...")
        neg = False
        if self.kw("not") and self.peek(1)[1] and self.peek(1)[1].lower() in ("in", "like"):
            self.take()
            neg = True
        if self.kw("in"):
            self.take()
            self.take("op", "(")
            if self.kw("select") or self.kw("with"):
                q = self.query()
                self.take("op", ")")
                return ("in_subq", neg, left, q)
            items = [self.expr()]
            while self.peek()[1] == ",":
                self.take()
                items.append(self.expr())
            self.take("op", ")")
            return ("in", neg, left, items)
        if self.kw("like"):
            self.take()
            return ("like", neg, left, self.add())
        if self.kw("between"):
            self.take()
            lo = self.add()
            self.take("id", "and")
            return ("and", ("cmp", ">=", left, lo), ("cmp", "<=", left, self.add()))
        return left

    def add(self):
        left = self.mul()
        while self.peek()[0] == "op" and self.peek()[1] in ("+", "-"):
            op = self.take()
            left = ("arith", op, left, self.mul())
        return left

    def mul(self):
        left = self.unary()
        while self.peek()[0] == "op" and self.peek()[1] in ("*", "/", "%"):
            op = self.take()
            left = ("arith", op, left, self.unary())
        return left

    def unary(self):
        if self.peek() == ("op", "-"):
            self.take()
            return ("neg", self.unary())
        return self.postfix()

    def postfix(self):
        node = self.primary()
        while True:
            k, v = self.peek()
            if v == "." and self.peek(1)[0] == "id":
                self.take()
                node = ("field", node, self.take("id").strip('"'))
            elif v == "[":
                self.take()
                idx = self.expr()
                if self.peek()[1] == ":":
                    self.take()
                    hi = self.expr()
                    self.take("op", "]")
                    node = ("slice", node, idx, hi)
                else:
                    self.take("op", "]")
                    node = ("index", node, idx)
            elif v == "::":
                self.take()
                node = ("cast", node, self.type_name())
            else:
                return node

    def type_name(self):
        name = self.take("id").upper()
        while self.peek()[1] in ("[", "("):
            close = "]" if self.take() == "[" else ")"
            depth = 1
            while depth:
                _, v = self.peek()
                if v is None:
                    raise Unsupported("type")
                self.take()
                depth += v in ("[", "(")
                depth -= v in ("]", ")")
            name += "[]" if close == "]" else ""
        return name

    def primary(self):
        k, v = self.peek()
        if k == "num":
            self.take()
            return ("lit", float(v) if re.search(r"[.eE]", v) else int(v))
        if k == "str":
            self.take()
            return ("lit", v[1:-1].replace("''", "'"))
        if v == "(":
            self.take()
            if self.kw("select") or self.kw("with"):
                q = self.query()
                self.take("op", ")")
                return ("subq", q)
            e = self.expr()
            self.take("op", ")")
            return e
        if v == "{":
            self.take()
            fields = []
            while self.peek()[1] != "}":
                key = self.take()
                key = key[1:-1] if key.startswith("'") else key
                self.take("op", ":")
                fields.append((key, self.expr()))
                if self.peek()[1] == ",":
                    self.take()
            self.take()
            return ("struct", fields)
        if v == "[":
            self.take()
            items = []
            while self.peek()[1] != "]":
                items.append(self.expr())
                if self.peek()[1] == ",":
                    self.take()
            self.take()
            return ("list", items)
        if v == "*":
            self.take()
            return ("star",)
        if k == "id":
            low = v.lower()
            if low in ("true", "false"):
                self.take()
                return ("lit", low == "true")
            if low == "null":
                self.take()
                return ("lit", None)
            if low == "cast" and self.peek(1)[1] == "(":
                self.take()
                self.take()
                e = self.expr()
                self.take("id", "as")
                t = self.type_name()
                self.take("op", ")")
                return ("cast", e, t)
            if low == "case":
                raise Unsupported("CASE")
            if self.peek(1)[1] == "(":
                self.take()
                self.take()
                args = []
                if self.kw("distinct"):
                    self.take()
                    low += "_distinct"
                while self.peek()[1] != ")":
                    if self.peek()[0] == "id" and self.peek(1)[1] == "->":
                        var = self.take()
                        self.take()
                        args.append(("lambda", var, self.expr()))
                    else:
                        args.append(self.expr())
                    if self.peek()[1] == ",":
                        self.take()
                self.take()
                if self.kw("filter"):
                    raise Unsupported("FILTER")
                if self.kw("over"):
                    self.take()
                    self.take("op", "(")
                    if self.kw("partition"):
                        raise Unsupported("PARTITION BY")
                    order = []
                    if self.kw("order"):
                        self.take()
                        self.take("id", "by")
                        while True:
                            node = self.expr()
                            desc = False
                            if self.kw("asc") or self.kw("desc"):
                                desc = self.take().lower() == "desc"
                            order.append((node, desc, ""))
                            if self.peek()[1] == ",":
                                self.take()
                                continue
                            break
                    self.take("op", ")")
                    if low not in ("lag", "lead", "row_number"):
                        raise Unsupported("window function " + low)
                    return ("window", low, args, order)
                return ("call", low, args)
            self.take()
            return ("col", v.strip('"'))
        raise Unsupported(f"primary at {v}")


CLAUSE_WORDS = ("from", "where", "group", "order", "limit", "having", "union", "except", "intersect", "join", "on", "as", "inner",
                "left", "right", "full", "cross", "natural", "using")


def _select(self):
    """SELECT items FROM relation [WHERE] [GROUP BY] [ORDER BY] [LIMIT] -> dict."""
    self.take("id", "select")
    q = {"items": [], "from": None, "where": None, "group": [], "order": [], "limit": None, "distinct": False}
    if self.kw("distinct"):
        self.take()
        q["distinct"] = True
    while True:
        node = self.expr()
        alias = None
        if self.kw("as"):
            self.take()
            alias = self.take("id").strip('"')
        elif self.peek()[0] == "id" and self.peek()[1].lower() not in CLAUSE_WORDS:
            alias = self.take("id").strip('"')
        q["items"].append((node, alias))
        if self.peek()[1] == ",":
            self.take()
            continue
        break
    q["joins"] = []
    if self.kw("from"):
        self.take()
        q["from"] = self.relation()
        q["alias"] = self.table_alias()
        while True:
            if self.peek()[1] == ",":
                self.take()
                rel = self.relation()
                q["joins"].append((rel, self.table_alias(), None))
                continue
            if self.kw("inner"):
                self.take()
            if self.kw("left") or self.kw("right") or self.kw("full") or self.kw("cross") or self.kw("natural"):
                raise Unsupported("outer / cross join")
            if not self.kw("join"):
                break
            self.take()
            rel = self.relation()
            alias = self.table_alias()
            if self.kw("using"):
                self.take()
                self.take("op", "(")
                cols = [self.take("id")]
                while self.peek()[1] == ",":
                    self.take()
                    cols.append(self.take("id"))
                self.take("op", ")")
                q["joins"].append((rel, alias, ("using", cols)))
            else:
                self.take("id", "on")
                q["joins"].append((rel, alias, self.expr()))
    if self.kw("where"):
        self.take()
        q["where"] = self.expr()
    if self.kw("group"):
        self.take()
        self.take("id", "by")
        q["group"].append(self.expr())
        while self.peek()[1] == ",":
            self.take()
            q["group"].append(self.expr())
    if self.kw("having"):
        self.take()
        q["having"] = self.expr()
    if self.kw("order"):
        self.take()
        self.take("id", "by")
        while True:
            node = self.expr()
            desc, nulls = False, ""
            if self.kw("asc") or self.kw("desc"):
                desc = self.take().lower() == "desc"
            if self.kw("nulls"):
                self.take()
                nulls = self.take().upper()
            q["order"].append((node, desc, nulls))
            if self.peek()[1] == ",":
                self.take()
                continue
            break
    if self.kw("limit"):
        self.take()
        q["limit"] = int(self.take("num"))
    return q


def _table_alias(self):
    if self.kw("as"):
        self.take()
        return self.take("id").strip('"')
    if self.peek()[0] == "id" and self.peek()[1].lower() not in CLAUSE_WORDS:
        return self.take("id").strip('"')
    return None


def _query(self):
    """[WITH name AS (query), ...] term {UNION [ALL] | EXCEPT | INTERSECT term}; term = SELECT ... | ( query )."""
    ctes = []
    if self.kw("with"):
        self.take()
        while True:
            name = self.take("id")
            self.take("id", "as")
            self.take("op", "(")
            ctes.append((name, self.query()))
            self.take("op", ")")
            if self.peek()[1] == ",":
                self.take()
                continue
            break

    def term():
        if self.peek()[1] == "(":
            self.take()
            q = self.query()
            self.take("op", ")")
            return q
        return self.select()

    node = term()
    while self.kw("union") or self.kw("except") or self.kw("intersect"):
        op = self.take().lower()
        keep_all = False
        if self.kw("all"):
            self.take()
            keep_all = True
        node = {"setop": op, "all": keep_all, "left": node, "right": term()}
    if ctes:
        node = {"with": ctes, "body": node}
    return node


def _relation(self):
    if self.peek()[1] == "(":
        self.take()
        if self.kw("describe"):
            self.take()
            rel = ("describe", self.query())
        else:
            rel = ("select", self.query())
        self.take("op", ")")
        return rel
    name = self.take("id")
    if self.peek()[1] != "(":
        return ("table", name)
    self.take()
    args, named = [], {}
    while self.peek()[1] != ")":
        if self.peek()[0] == "id" and self.peek(1)[1] == ":=":
            key = self.take()
            self.take()
            named[key] = self.expr()
        else:
            args.append(self.expr())
        if self.peek()[1] == ",":
            self.take()
    self.take()
    return ("call", name.lower(), args, named)


Parser.select = _select
Parser.relation = _relation
Parser.query = _query
Parser.table_alias = _table_alias


def parse_sql(sql):
    p = Parser(sql.strip().rstrip(";"))
    q = p.query()
    if p.peek()[1] == ";":
        p.take()
    if not p.done():
        raise Unsupported("text behind the statement: " + str(p.peek()[1]))
    return q


def walk_selects(q, fn):
    """fn(select dict) for the statement and every subquery in it."""
    if "setop" in q:
        walk_selects(q["left"], fn)
        walk_selects(q["right"], fn)
        return
    if "with" in q:
        for _, sub in q["with"]:
            walk_selects(sub, fn)
        walk_selects(q["body"], fn)
        return
    fn(q)

    def in_node(n):
        if isinstance(n, tuple):
            if n[0] == "subq":
                walk_selects(n[1], fn)
            for x in n[1:]:
                in_node(x)
        elif isinstance(n, list):
            for x in n:
                in_node(x)
        elif isinstance(n, dict):
            for x in n.values():
                in_node(x)

    for node, _ in q["items"]:
        in_node(node)
    in_node(q["where"])
    in_node(q["group"])
    for node, _, _ in q["order"]:
        in_node(node)
    in_node(q.get("having"))
    for rel in [q["from"]] + [j[0] for j in q["joins"]]:
        if rel and rel[0] in ("select", "describe"):
            walk_selects(rel[1], fn)
        if rel and rel[0] == "call":
            in_node(rel[2])
            in_node(rel[3])
    for _, _, on in q["joins"]:
        if isinstance(on, tuple) and on[0] != "using":
            in_node(on)


def check_select(q, functions):
    """Raises Unsupported unless every relation is one of `functions` and every function call is known."""
    calls = []

    cte_names = set()

    def names_of(node):
        if "with" in node:
            cte_names.update(n.lower() for n, _ in node["with"])
            for _, sub in node["with"]:
                names_of(sub)
            names_of(node["body"])
        elif "setop" in node:
            names_of(node["left"])
            names_of(node["right"])

    names_of(q)

    def one(sel):
        for rel in [sel["from"]] + [j[0] for j in sel["joins"]]:
            if rel and rel[0] == "table" and rel[1].lower() not in cte_names:
                raise Unsupported("table " + rel[1])
            if rel and rel[0] == "call":
                if rel[1] not in functions:
                    raise Unsupported("relation " + rel[1])
                calls.append(rel)
                for node in list(rel[2]) + list(rel[3].values()):
                    check(node)
        for _, _, on in sel["joins"]:
            if isinstance(on, tuple) and on[0] != "using":
                check(on)
        check(sel.get("having"))
        for node, _ in sel["items"]:
            check(node)
        check(sel["where"])
        for g in sel["group"]:
            check(g)
        for node, _, _ in sel["order"]:
            check(node)
        aggs = [has_aggregate(n) for n, _ in sel["items"]]
        if any(aggs) and not all(aggs) and not sel["group"]:
            raise Unsupported("aggregates next to plain columns")

    walk_selects(q, one)
    if not calls:
        raise Unsupported("no call of this path")
    return calls


def literal(node, env=None):
    """The Python value of a literal argument expression (lists, structs, casts of them; session variables from env)."""
    return ev(node, env or {}, env or {})


def item_name(node, alias):
    if alias:
        return alias
    if node[0] == "col":
        return node[1]
    if node[0] == "field":
        return node[2]
    if node[0] == "call":
        return node[1] + "(...)"
    return "?"


def run_select(q, provider, ctes=None):
    """-> (names, types, rows).  provider(function, args, named) -> object with .names, .types, .rows."""
    ctes = dict(ctes or {})
    if "with" in q:
        for name, sub in q["with"]:
            ctes[name.lower()] = run_select(sub, provider, ctes)
        return run_select(q["body"], provider, ctes)
    if "setop" in q:
        ln, lt, lrows = run_select(q["left"], provider, ctes)
        _, _, rrows = run_select(q["right"], provider, ctes)
        lk, rk = [_hashable(r) for r in lrows], [_hashable(r) for r in rrows]
        if q["setop"] == "union":
            rows = list(lrows) + list(rrows)
            if not q["all"]:
                rows = _dedupe(rows)
        elif q["setop"] == "except":
            gone = set(rk)
            rows = _dedupe([r for r, k in zip(lrows, lk) if k not in gone])
        else:
            keep = set(rk)
            rows = _dedupe([r for r, k in zip(lrows, lk) if k in keep])
        return ln, lt, rows

    def relation(rel):
        if rel[0] == "call":
            env = {"__variables__": getattr(provider, "variables", None)}
            r = provider(rel[1], [literal(a, env) for a in rel[2]], {k: literal(v, env) for k, v in rel[3].items()})
            return list(r.names), list(r.types), list(r.rows)
        if rel[0] == "table":
            n, t, rows = ctes[rel[1].lower()]
            return list(n), list(t), list(rows)
        if rel[0] == "describe":
            n, t, _ = run_select(rel[1], provider, ctes)
            return ["column_name", "column_type"], ["VARCHAR", "VARCHAR"], list(zip(n, t))
        return run_select(rel[1], provider, ctes)

    extra = {"__provider__": provider, "__ctes__": ctes, "__variables__": getattr(provider, "variables", None),
             "__aliases__": {a.lower(): n for n, a in q["items"] if a and not has_aggregate(n)}}
    rel = q["from"]
    if rel is None:
        names, types, dicts = [], [], [dict(extra)]
    else:
        names, types, rows = relation(rel)
        dicts = []
        first_alias = q.get("alias") or (rel[1] if rel[0] == "table" else None)
        for r in rows:
            d = dict(zip(names, r))
            if first_alias:
                d[first_alias] = dict(zip(names, r))
            d.update(extra)
            dicts.append(d)
        for jrel, jalias, on in q["joins"]:
            jalias = jalias or (jrel[1] if jrel[0] == "table" else None)
            jn, jt, jrows = relation(jrel)
            jd = [dict(zip(jn, r)) for r in jrows]
            joined = []
            key_l = key_r = None
            if isinstance(on, tuple) and on[0] == "using":
                key_l = [("col", c) for c in on[1]]
                key_r = key_l
            elif isinstance(on, tuple) and on[0] == "cmp" and on[1] == "=":
                key_l, key_r = [on[2]], [on[3]]
            tmap0 = dict(zip(names + jn, types + jt))
            tmap0.update(extra)

            def right_row(d):
                out = dict(d)
                if jalias:
                    out = {jalias: dict(d)}
                    out.update({k: v for k, v in d.items()})
                return out

            if key_l is not None:
                # an equality join: hash one side (whichever side of the = names the joined relation)
                def try_keys(dl, dr, kl, kr):
                    return tuple(_hashable(ev(k, dl, tmap0)) for k in kl), tuple(_hashable(ev(k, dr, tmap0)) for k in kr)
                index = None
                for kl, kr in ((key_l, key_r), (key_r, key_l)):
                    try:
                        index = {}
                        for d in jd:
                            index.setdefault(tuple(_hashable(ev(k, right_row(d), tmap0)) for k in kr), []).append(d)
                        probe = kl
                        if dicts:
                            tuple(_hashable(ev(k, dicts[0], tmap0)) for k in probe)
                        break
                    except (KeyError, TypeError):
                        index = None
                if index is None:
                    raise Unsupported("join keys")
                for d in dicts:
                    for m in index.get(tuple(_hashable(ev(k, d, tmap0)) for k in probe), []):
                        merged = dict(right_row(m))
                        merged.update(d)  # the left side wins an unqualified name clash, as the first relation does
                        joined.append(merged)
            else:
                if len(dicts) * len(jd) > 4_000_000:
                    raise Unsupported("join too large for the nested loop")
                for d in dicts:
                    for m in jd:
                        merged = dict(right_row(m))
                        merged.update(d)
                        if on is None or ev(on, merged, tmap0) is True:
                            joined.append(merged)
            dicts = joined
            names, types = names + [n for n in jn if n not in names], types + [t for n, t in zip(jn, jt) if n not in names]
    tmap = dict(zip(names, types))
    tmap.update(extra)
    if q["where"] is not None:
        dicts = [d for d in dicts if ev(q["where"], d, tmap) is True]
    items = q["items"]
    out_names = []
    for node, alias in items:
        out_names.extend(names if node == ("star",) else [item_name(node, alias)])
    # window functions (no PARTITION BY): a value per row of the filtered relation, in the window's own order
    windows = []

    def find_windows(n):
        if isinstance(n, tuple):
            if n and n[0] == "window":
                windows.append(n)
                return
            for x in n[1:]:
                find_windows(x)
        elif isinstance(n, list):
            for x in n:
                find_windows(x)

    for node, _ in items:
        find_windows(node)
    for d in dicts:
        d["__win__"] = {}
    for wnode in windows:
        order = sorted(range(len(dicts)), key=lambda i: _order_key(wnode[3], dicts[i], tmap)) if wnode[3] else list(range(len(dicts)))
        for pos, i in enumerate(order):
            if wnode[1] == "row_number":
                v = pos + 1
            else:
                j = pos - 1 if wnode[1] == "lag" else pos + 1
                v = ev(wnode[2][0], dicts[order[j]], tmap) if 0 <= j < len(order) else None
            dicts[i]["__win__"][id(wnode)] = v
    is_agg = any(has_aggregate(n) for n, _ in items)
    if q["group"]:
        groups = {}
        for d in dicts:
            key = tuple(_hashable(ev(g, d, tmap)) for g in q["group"])
            groups.setdefault(key, []).append(d)
        out = []
        for key, members in groups.items():
            row = tuple(agg(n, members, tmap) if has_aggregate(n) else ev(n, members[0], tmap) for n, _ in items)
            if q.get("having") is not None:
                named = dict(members[0])
                named.update({a: v for (_, a), v in zip(items, row) if a})
                if agg(q["having"], [named] if not has_aggregate(q["having"]) else members, tmap) is not True:
                    continue
            out.append((members[0], row, members))
        if q["order"]:
            out.sort(key=lambda t: _order_key(q["order"], t[0], tmap, dict(zip(out_names, t[1])), None, t[2]))
        rows_out = [t[1] for t in out]
    elif is_agg:
        rows_out = [tuple(agg(n, dicts, tmap) for n, _ in items)]
    else:
        if q["order"]:
            dicts.sort(key=lambda d: _order_key(q["order"], d, tmap, None, items))
        rows_out = []
        for d in dicts:
            row = []
            for n, _ in items:
                if n == ("star",):
                    row.extend(d[c] for c in names)
                else:
                    row.append(ev(n, d, tmap))
            rows_out.append(tuple(row))
        if q["distinct"]:
            rows_out = _dedupe(rows_out)
    if q["limit"] is not None:
        rows_out = rows_out[:q["limit"]]
    out_types = []
    for node, alias in items:
        if node == ("star",):
            out_types.extend(types)
        else:
            try:
                out_types.append(types_of(node, tmap))
            except (Unsupported, KeyError):
                out_types.append("?")
    return out_names, out_types, rows_out


def _dedupe(rows):
    seen, out = set(), []
    for r in rows:
        k = _hashable(r)
        if k not in seen:
            seen.add(k)
            out.append(r)
    return out


def _hashable(v):
    if isinstance(v, (list, tuple)):
        return tuple(_hashable(x) for x in v)
    if isinstance(v, dict):
        return tuple((k, _hashable(x)) for k, x in v.items())
    return v


def _order_key(order, d, tmap, computed=None, items=None, members=None):
    out = []
    for node, desc, nulls in order:
        if computed is not None and node[0] == "col" and node[1] in computed:
            v = computed[node[1]]
        elif members is not None and has_aggregate(node):
            v = agg(node, members, tmap)
        else:
            if items and node[0] == "col" and not any(k.lower() == node[1].lower() for k in d):
                hit = next((n for n, a in items if a and a.lower() == node[1].lower()), None)
                v = ev(hit, d, tmap) if hit else ev(node, d, tmap)
            elif node[0] == "lit" and isinstance(node[1], int) and items:
                v = ev(items[node[1] - 1][0], d, tmap)  # ORDER BY 1
            else:
                v = ev(node, d, tmap)
        null_last = (nulls == "LAST") if nulls else True
        out.append((v is None) == null_last)
        out.append(Rev(v) if desc else Fwd(v))
    return tuple(out)


def split_top(text):
    """Select-list / order-by items: split at top-level commas."""
    items, depth, cur, quote = [], 0, [], False
    for ch in text:
        if ch == "'":
            quote = not quote
        if not quote:
            depth += ch in "([{"
            depth -= ch in ")]}"
            if ch == "," and depth == 0:
                items.append("".join(cur).strip())
                cur = []
                continue
        cur.append(ch)
    items.append("".join(cur).strip())
    return items


def has_aggregate(node):
    if not isinstance(node, tuple):
        return False
    if node[0] == "call" and node[1] in AGGREGATES:
        return True
    return any(has_aggregate(x) if isinstance(x, tuple) else any(has_aggregate(y) for y in x) if isinstance(x, list) else False
               for x in node[1:])


SCALARS = {"round", "abs", "typeof", "len", "length", "array_length", "list_sum", "list_count", "list_contains", "lower", "upper",
           "floor", "ceil", "coalesce", "isnan", "list_min", "list_max", "array_extract", "list_extract", "struct_extract", "sqrt",
           "ln", "exp", "greatest", "least", "range", "list_transform", "concat", "concat_ws", "getvariable"}


def check(node):
    if not isinstance(node, tuple):
        return
    if node[0] == "call" and node[1] not in SCALARS and node[1] not in AGGREGATES:
        raise Unsupported("function " + node[1])
    for x in node[1:]:
        if isinstance(x, tuple):
            check(x)
        elif isinstance(x, list):
            for y in x:
                check(y)


def compile_query(select, where, order):
    items = []
    for text in split_top(select):
        text = re.sub(r"(?i)\s+AS\s+\"?[A-Za-z_][A-Za-z_0-9]*\"?\s*$", "", text)
        p = Parser(text)
        node = p.expr()
        if not p.done():  # an alias without AS
            k, v = p.peek()
            if k == "id" and p.i == len(p.t) - 1:
                pass
            else:
                raise Unsupported("select item: " + text[:30])
        check(node)
        items.append(node)
    aggs = [has_aggregate(n) for n in items]
    if any(aggs) and not all(aggs):
        raise Unsupported("aggregates next to plain columns")
    w = None
    if where:
        p = Parser(where)
        w = p.expr()
        if not p.done():
            raise Unsupported("where")
        check(w)
    o = []
    if order:
        for text in split_top(order):
            m = re.match(r"(?is)^(.*?)(?:\s+(ASC|DESC))?(?:\s+NULLS\s+(FIRST|LAST))?$", text)
            p = Parser(m.group(1))
            node = p.expr()
            if not p.done():
                raise Unsupported("order by")
            check(node)
            o.append((node, (m.group(2) or "ASC").upper() == "DESC", (m.group(3) or "").upper()))
    return items, w, o, any(aggs)


# ---- evaluation ----

def type_of(v, declared=None):
    return declared


def ev(node, row, types):
    kind = node[0]
    if kind == "lit":
        return node[1]
    if kind == "col":
        for name in (node[1], node[1].upper(), node[1].lower()):
            if name in row:
                return row[name]
        for name in row:
            if name.lower() == node[1].lower():
                return row[name]
        alias = (row.get("__aliases__") or {}).get(node[1].lower())
        if alias is not None and alias != node:
            return ev(alias, row, types)
        raise KeyError(node[1])
    if kind == "field":
        base = ev(node[1], row, types)
        return None if base is None else base[node[2]]
    if kind == "index":
        base, idx = ev(node[1], row, types), ev(node[2], row, types)
        if base is None or idx is None:
            return None
        if isinstance(base, dict):
            return base[idx]
        return base[idx - 1] if 1 <= idx <= len(base) else None
    if kind == "list":
        return [ev(x, row, types) for x in node[1]]
    if kind == "struct":
        return {k: ev(x, row, types) for k, x in node[1]}
    if kind == "window":
        return row["__win__"][id(node)]
    if kind == "slice":
        base, lo, hi = ev(node[1], row, types), ev(node[2], row, types), ev(node[3], row, types)
        return None if base is None else list(base[lo - 1:hi])
    if kind == "in_subq":
        a = ev(node[2], row, types)
        if a is None:
            return None
        _, _, rows = run_select(node[3], row.get("__provider__") or types.get("__provider__"),
                                row.get("__ctes__") or types.get("__ctes__"))
        return (_hashable(a) in {_hashable(r[0]) for r in rows}) != node[1]
    if kind == "subq":
        _, _, rows = run_select(node[1], row.get("__provider__") or types.get("__provider__"),
                                row.get("__ctes__") or types.get("__ctes__"))
        return rows[0][0] if rows else None
    if kind == "neg":
        v = ev(node[1], row, types)
        return None if v is None else -v
    if kind == "arith":
        a, b = ev(node[2], row, types), ev(node[3], row, types)
        if a is None or b is None:
            return None
        if node[1] == "+":
            return a + b
        if node[1] == "-":
            return a - b
        if node[1] == "*":
            return a * b
        if node[1] == "/":
            return a / b if b else None
        return a % b
    if kind == "cmp":
        a, b = ev(node[2], row, types), ev(node[3], row, types)
        if a is None or b is None:
            return None
        if isinstance(a, tuple):
            a = list(a)
        op = node[1]
        return (a == b if op == "=" else a != b if op in ("<>", "!=") else a < b if op == "<" else a <= b if op == "<=" else
                a > b if op == ">" else a >= b)
    if kind == "distinct":
        a, b = ev(node[2], row, types), ev(node[3], row, types)
        return (a != b) == node[1]
    if kind == "isnull":
        return (ev(node[2], row, types) is None) != node[1]
    if kind == "in":
        a = ev(node[2], row, types)
        if a is None:
            return None
        hit = a in [ev(x, row, types) for x in node[3]]
        return hit != node[1]
    if kind == "like":
        a, pat = ev(node[2], row, types), ev(node[3], row, types)
        if a is None:
            return None
        rx = "^" + "".join(".*" if c == "%" else "." if c == "_" else re.escape(c) for c in pat) + "$"
        return (re.match(rx, a) is not None) != node[1]
    if kind == "and":
        a, b = ev(node[1], row, types), ev(node[2], row, types)
        return False if a is False or b is False else None if a is None or b is None else True
    if kind == "or":
        a, b = ev(node[1], row, types), ev(node[2], row, types)
        return True if a is True or b is True else None if a is None or b is None else False
    if kind == "not":
        a = ev(node[1], row, types)
        return None if a is None else not a
    if kind == "cast":
        v, t = ev(node[1], row, types), node[2]
        if v is None:
            return None
        if t in ("VARCHAR", "TEXT", "STRING"):
            return duck_str(v)
        if t in ("INTEGER", "BIGINT", "INT", "UINTEGER", "SMALLINT", "TINYINT", "UBIGINT", "HUGEINT"):
            return int(round(v)) if isinstance(v, float) else int(v)
        if t in ("DOUBLE", "FLOAT", "REAL"):
            return float(v)
        if t == "BOOLEAN":
            return bool(v)
        if t.endswith("[]"):  # a list / array type: the value is the list
            return list(v)
        raise Unsupported("cast to " + t)
    if kind == "call":
        name = node[1]
        if name == "typeof":
            return types_of(node[2][0], types)
        if name == "list_transform":
            base, lam = ev(node[2][0], row, types), node[2][1]
            out = []
            for item in base:
                scope = dict(row)
                scope[lam[1]] = item
                out.append(ev(lam[2], scope, types))
            return out
        if name == "getvariable":
            return (row.get("__variables__") or types.get("__variables__") or {})[ev(node[2][0], row, types)]
        a = [ev(x, row, types) for x in node[2]]
        if name == "range":
            return list(range(*a))
        if name == "concat":
            return "".join(duck_str(x) for x in a if x is not None)
        if name == "concat_ws":
            return a[0].join(duck_str(x) for x in a[1:] if x is not None)
        if name == "coalesce":
            return next((x for x in a if x is not None), None)
        if a and a[0] is None:
            return None
        if name == "round":
            nd = a[1] if len(a) > 1 else 0
            q = 10 ** nd
            v = a[0] * q
            r = math.floor(abs(v) + 0.5) * (1 if v >= 0 else -1) / q  # half away from zero, as DuckDB rounds
            return r if nd > 0 or isinstance(a[0], float) else int(r)
        if name == "abs":
            return abs(a[0])
        if name in ("len", "length", "array_length", "list_count"):
            return len(a[0]) if name != "list_count" else sum(x is not None for x in a[0])
        if name == "list_sum":
            vals = [x for x in a[0] if x is not None]
            return sum(vals) if vals else None
        if name == "list_min":
            vals = [x for x in a[0] if x is not None]
            return min(vals) if vals else None
        if name == "list_max":
            vals = [x for x in a[0] if x is not None]
            return max(vals) if vals else None
        if name == "list_contains":
            return a[1] in a[0]
        if name in ("array_extract", "list_extract"):
            return a[0][a[1] - 1] if 1 <= a[1] <= len(a[0]) else None
        if name == "struct_extract":
            return a[0][a[1]]
        if name == "lower":
            return a[0].lower()
        if name == "upper":
            return a[0].upper()
        if name == "floor":
            return math.floor(a[0])
        if name == "ceil":
            return math.ceil(a[0])
        if name == "isnan":
            return isinstance(a[0], float) and math.isnan(a[0])
        if name == "sqrt":
            return math.sqrt(a[0])
        if name == "ln":
            return math.log(a[0])
        if name == "exp":
            return math.exp(a[0])
        if name == "greatest":
            return max(a)
        if name == "least":
            return min(a)
    raise Unsupported("eval " + kind)


def types_of(node, types):
    """typeof(expr) for a column, a struct field or a list element of one."""
    if node[0] == "col":
        for name, t in types.items():
            if name.lower() == node[1].lower():
                return t
        raise KeyError(node[1])
    if node[0] == "field":
        base = types_of(node[1], types)
        m = re.search(r"[(,]\s*\"?" + re.escape(node[2]) + r"\"?\s+([A-Z]+(?:\([^()]*\))?(?:\[\d*\])*)", base)
        if m:
            return m.group(1)
    if node[0] == "index":
        base = types_of(node[1], types)
        m = re.match(r"^(.*)\[\d*\]$", base)
        if m:
            return m.group(1)
    raise Unsupported("typeof of an expression")


def agg(node, rows, types):
    if node[0] == "call" and node[1] in AGGREGATES:
        name, args = node[1], node[2]
        if name == "count":
            if not args or args[0] == ("star",):
                return len(rows)
            return sum(ev(args[0], r, types) is not None for r in rows)
        vals = [ev(args[0], r, types) for r in rows]
        if name == "count_distinct":
            return len({_hashable(v) for v in vals if v is not None})
        if name == "list":
            return vals
        vals = [v for v in vals if v is not None]
        if name in ("first", "any_value"):
            return vals[0] if vals else None
        if not vals:
            return None
        if name == "sum":
            return math.fsum(vals) if any(isinstance(v, float) for v in vals) else sum(vals)
        if name == "min":
            return min(vals)
        if name == "max":
            return max(vals)
        if name == "avg":
            return math.fsum(vals) / len(vals)
        if name == "bool_and":
            return all(vals)
        if name == "bool_or":
            return any(vals)
    if node[0] in ("lit",):
        return node[1]
    # an expression over aggregates: evaluate the aggregates first
    if node[0] == "arith":
        a, b = agg(node[2], rows, types), agg(node[3], rows, types)
        if a is None or b is None:
            return None
        return ev(("arith", node[1], ("lit", a), ("lit", b)), {}, types)
    if node[0] == "cmp":
        return ev(("cmp", node[1], ("lit", agg(node[2], rows, types)), ("lit", agg(node[3], rows, types))), {}, types)
    if node[0] == "cast":
        return ev(("cast", ("lit", agg(node[1], rows, types)), node[2]), {}, types)
    if node[0] == "call":
        return ev(("call", node[1], [("lit", agg(x, rows, types)) for x in node[2]]), {}, types)
    if node[0] == "neg":
        v = agg(node[1], rows, types)
        return None if v is None else -v
    if node[0] in ("and", "or"):
        return ev((node[0], ("lit", agg(node[1], rows, types)), ("lit", agg(node[2], rows, types))), {}, types)
    if node[0] in ("col", "field", "index", "subq"):
        return ev(node, rows[0], types) if rows else None
    raise Unsupported("aggregate expression")


def run(compiled, rows, names, types, limit=None):
    """rows: tuples in `names` order -> list of result tuples."""
    items, where, order, is_agg = compiled
    tmap = dict(zip(names, types))
    dicts = [dict(zip(names, r)) for r in rows]
    if where is not None:
        dicts = [d for d in dicts if ev(where, d, tmap) is True]
    if order:
        def key(d):
            out = []
            for node, desc, nulls in order:
                v = ev(node, d, tmap)
                null_last = (nulls == "LAST") if nulls else True  # DuckDB: NULLS LAST by default for ASC and DESC alike
                out.append((v is None) == null_last)
                out.append(Rev(v) if desc else Fwd(v))
            return tuple(out)
        dicts.sort(key=key)
    if is_agg:
        return [tuple(agg(n, dicts, tmap) for n in items)]
    if limit is not None:
        dicts = dicts[:limit]
    out = []
    for d in dicts:
        row = []
        for n in items:
            if n == ("star",):
                row.extend(d[c] for c in names)
            else:
                row.append(ev(n, d, tmap))
        out.append(tuple(row))
    return out


class Fwd:
    def __init__(self, v):
        self.v = v

    def __lt__(self, o):
        return False if self.v is None or o.v is None else self.v < o.v

    def __eq__(self, o):
        return self.v == o.v


class Rev(Fwd):
    def __lt__(self, o):
        return False if self.v is None or o.v is None else self.v > o.v


def referenced_columns(compiled):
    """Column names the query reads, or None when it reads every column (`*`)."""
    items, where, order, _ = compiled
    cols, star = set(), False

    def walk(n):
        nonlocal star
        if isinstance(n, tuple):
            if n[0] == "col":
                cols.add(n[1])
            if n == ("star",):
                star = True
            for x in n[1:]:
                walk(x)
        elif isinstance(n, list):
            for x in n:
                walk(x)

    for n in items:
        if n == ("star",):
            star = True
        elif n[0] == "call" and n[1] == "count" and (not n[2] or n[2][0] == ("star",)):
            continue
        else:
            walk(n)
    walk(where)
    for node, _, _ in order:
        walk(node)
    return None if star else cols


def duck_str(v):
    """DuckDB's text rendering of a value (what a T column of a sqllogictest shows)."""
    if v is None:
        return "NULL"
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, float):
        if math.isnan(v):
            return "nan"
        if math.isinf(v):
            return "inf" if v > 0 else "-inf"
        if v == int(v) and abs(v) < 1e15:
            return f"{v:.1f}"
        return repr(v)
    if isinstance(v, (list, tuple)):
        return "[" + ", ".join(duck_str(x) for x in v) + "]"
    if isinstance(v, dict):
        return "{" + ", ".join(f"'{k}': {duck_str(x)}" for k, x in v.items()) + "}"
    return str(v)


NUMBER = re.compile(r"-?\d+\.\d+(?:[eE][-+]?\d+)?|-?\d+[eE][-+]?\d+|-?\d+")


def same_text(got, want, rel=1e-9, abs_tol=1e-12):
    """Equal texts, numbers inside compared with a tolerance (a double's last digits depend on summation order)."""
    if got == want:
        return True
    g_parts, w_parts = NUMBER.split(got), NUMBER.split(want)
    if g_parts != w_parts:
        return False
    g_num, w_num = NUMBER.findall(got), NUMBER.findall(want)
    return all(math.isclose(float(a), float(b), rel_tol=rel, abs_tol=abs_tol) for a, b in zip(g_num, w_num))


def matches(value, want, type_char):
    """One cell against the expected text of the reference's test."""
    if want == "NULL":
        return value is None
    if value is None:
        return False
    if type_char == "I":
        if isinstance(value, bool):
            return want in ("1", "true") if value else want in ("0", "false")
        try:
            if re.fullmatch(r"-?\d+", want):
                return int(round(float(value))) == int(want) and (not isinstance(value, float) or abs(value - round(value)) < 1e-9)
            return math.isclose(float(value), float(want), rel_tol=1e-6, abs_tol=1e-9)  # (a real under an I: compared as one)
        except (TypeError, ValueError):
            return same_text(duck_str(value), want)
    if type_char == "R":
        try:
            if isinstance(value, bool):
                return False
            w = float(want)
            v = float(value)
            return (math.isnan(v) and math.isnan(w)) or math.isclose(v, w, rel_tol=1e-6, abs_tol=1e-9)
        except (TypeError, ValueError):
            return same_text(duck_str(value), want)
    text = duck_str(value)
    return same_text(text, want) or (want == "(empty)" and text == "")

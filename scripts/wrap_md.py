#!/usr/bin/env python3
"""Re-flow the project's Markdown documents to at most WIDTH columns so that their diffs are reviewable.

  * every paragraph and list item is re-flowed as a whole (its lines joined, then broken at spaces); continuation lines
    are indented under the text of the list item (or under the paragraph's own indent), which Markdown reads as the same
    paragraph;
  * a table with a row longer than WIDTH cannot be wrapped (a row is one line), so it becomes a list: one item per row
    headed by its first cell, one sub-item per further column labelled with the column's header;
  * fenced code blocks, headings and tables that fit are left alone.

usage: wrap_md.py FILE [FILE ...]      (rewrites the files in place; running it twice changes nothing)
"""
import re
import sys

WIDTH = 120
LIST = re.compile(r"^(\s*)([*+-]|\d+[.)])(\s+)")


def width(s):
    """columns as the bluntest tool counts them: bytes of the UTF-8 encoding (an arrow or a Greek letter is 2-3)"""
    return len(s.encode("utf-8"))


def greedy(words, first, hang):
    out, cur = [], first
    fresh = True
    for w in words:
        if fresh:
            cur += w
            fresh = False
        elif width(cur) + 1 + width(w) <= WIDTH:
            cur += " " + w
        else:
            # a continuation line must not begin with something Markdown reads as a block start (a list marker, a heading,
            # a table row, a quote): the previous word comes down with it
            def risky(x):
                return x in ("-", "*", "+", ">") or re.match(r"^\d+[.)]$", x) is not None or x[:1] in ("#", "|", ">")

            lead = len(first) if not out else len(hang)
            down = w
            while risky(down.split(" ", 1)[0]) and " " in cur[lead:]:
                cur, last = cur.rsplit(" ", 1)
                down = last + " " + down
            out.append(cur)
            cur = hang + down
    out.append(cur)
    return out


def wrap_line(line):
    if width(line) <= WIDTH:
        return [line]
    m = LIST.match(line)
    if m:
        first = m.group(0)
        rest = line[len(first):]
        hang = " " * len(first)
    else:
        first = re.match(r"^\s*", line).group(0)
        rest = line[len(first):]
        hang = first
    words = rest.split(" ")
    words = [w for k, w in enumerate(words) if w != "" or k == 0]
    return greedy(words, first, hang) or [line]


def cells(row):
    row = row.strip()
    if row.startswith("|"):
        row = row[1:]
    if row.endswith("|"):
        row = row[:-1]
    # (a `|` inside a code span or escaped stays part of its cell)
    parts, cur, in_code, i = [], "", False, 0
    while i < len(row):
        ch = row[i]
        if ch == "`":
            in_code = not in_code
        if ch == "\\" and i + 1 < len(row) and row[i + 1] == "|":
            cur += "|"
            i += 2
            continue
        if ch == "|" and not in_code:
            parts.append(cur.strip())
            cur = ""
        else:
            cur += ch
        i += 1
    parts.append(cur.strip())
    return parts


def table_to_list(rows, indent):
    head = cells(rows[0])
    body = [cells(r) for r in rows[2:]]
    out = []
    for r in body:
        if not any(r):
            continue
        label = r[0] if r[0] else "(no label)"
        if not (label.startswith("**") and label.endswith("**")):
            label = "**%s**" % label
        prefix = "%s (%s)" % (label, head[0]) if head and head[0] and head[0] not in ("", "#") and len(body) > 0 and len(head[0]) < 40 \
            and head[0].lower() not in label.lower() else label
        out += wrap_line("%s* %s" % (indent, prefix))
        for h, c in zip(head[1:], r[1:]):
            if c == "":
                continue
            text = "%s: %s" % (h, c) if h else c
            out += wrap_line("%s  * %s" % (indent, text))
    return out


def reflow(text):
    lines = text.split("\n")
    out, i, fence = [], 0, False
    while i < len(lines):
        line = lines[i]
        if line.lstrip().startswith("```"):
            fence = not fence
            out.append(line)
            i += 1
            continue
        if fence or line.lstrip().startswith("#"):
            out.append(line)
            i += 1
            continue
        if line.lstrip().startswith("|"):
            j = i
            while j < len(lines) and lines[j].lstrip().startswith("|"):
                j += 1
            rows = lines[i:j]
            is_table = len(rows) >= 2 and re.match(r"^\s*\|?\s*:?-{2,}", rows[1]) is not None
            if is_table and max(width(r) for r in rows) > WIDTH:
                indent = re.match(r"^\s*", rows[0]).group(0)
                out += table_to_list(rows, indent)
            else:
                out += rows
            i = j
            continue
        if line.strip() == "":
            out.append(line)
            i += 1
            continue
        # a paragraph or list item: this line and the plain lines that continue it, re-flowed as one
        j = i + 1
        hard_break = line.endswith("  ")
        while (not hard_break and j < len(lines) and lines[j].strip() != "" and not LIST.match(lines[j])
               and not lines[j].lstrip().startswith(("#", "|", "```", ">"))):
            hard_break = lines[j].endswith("  ")
            j += 1
        m = LIST.match(line)
        first = m.group(0) if m else re.match(r"^\s*", line).group(0)
        hang = " " * len(first) if m else first
        text = " ".join([line[len(first):].strip()] + [ln.strip() for ln in lines[i + 1:j]])
        words = [w for w in text.split(" ") if w != ""]
        out += greedy(words, first, hang) if words else [line]
        i = j
    return "\n".join(out)


def main():
    for path in sys.argv[1:]:
        src = open(path).read()
        dst = reflow(src)
        if dst != src:
            open(path, "w").write(dst)
        long_left = sum(1 for ln in dst.split("\n") if width(ln) > WIDTH)
        print("%s: %d -> %d lines, %d still longer than %d" % (path, src.count("\n") + 1, dst.count("\n") + 1, long_left,
                                                               WIDTH))


if __name__ == "__main__":
    main()

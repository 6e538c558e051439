#!/usr/bin/env python3
"""TypeScript -> JavaScript type stripper (tooling; contains no reference code).

The reference (paulmillr/noble-bls12-381) is two TypeScript files.  This image has Node 12
but no tsc/ts-node/esbuild, so to *run the reference itself* (for golden fixtures and for
pinning oracle/) we erase the TypeScript-only syntax with a small scanner and write ES
modules to a scratch directory OUTSIDE the repo (default /tmp/nbls_ref).  The stripped
output is never committed and never travels to the GPU box; only the vectors it produces do
(tests/golden/, written by tools/gen_golden.py).

Usage: python tools/strip_ts.py [/root/reference] [/tmp/nbls_ref]
"""
import re
import sys
import os

OPEN = {'(': ')', '[': ']', '{': '}'}
KEYWORDS = {'if', 'for', 'while', 'switch', 'catch', 'return', 'typeof', 'new', 'throw', 'await', 'function', 'super'}


def skip_ws(s, i):
    while i < len(s) and s[i] in ' \t\r\n':
        i += 1
    return i


def skip_string(s, i):
    q = s[i]
    i += 1
    while i < len(s):
        c = s[i]
        if c == '\\':
            i += 2
            continue
        if q == '`' and c == '$' and s[i + 1:i + 2] == '{':
            i = match_close(s, i + 1) + 1
            continue
        if c == q:
            return i + 1
        i += 1
    return i


def match_close(s, i):
    """s[i] is an opening bracket; return index of its matching closer (skips strings/comments)."""
    stack = [OPEN[s[i]]]
    i += 1
    while i < len(s):
        c = s[i]
        if c in '"\'`':
            i = skip_string(s, i)
            continue
        if c == '/' and s[i + 1:i + 2] == '/':
            i = s.index('\n', i)
            continue
        if c == '/' and s[i + 1:i + 2] == '*':
            i = s.index('*/', i) + 2
            continue
        if c in OPEN:
            stack.append(OPEN[c])
        elif c in ')]}':
            assert stack and stack[-1] == c, (c, s[max(0, i - 80):i + 20])
            stack.pop()
            if not stack:
                return i
        i += 1
    raise ValueError('unbalanced')


def match_angle(s, i):
    """s[i] == '<' opening a generic argument list; return index of matching '>'."""
    depth = 0
    while i < len(s):
        c = s[i]
        if c == '<':
            depth += 1
        elif c == '>' and s[i - 1] != '=':
            depth -= 1
            if depth == 0:
                return i
        elif c in OPEN:
            i = match_close(s, i)
        i += 1
    raise ValueError('unbalanced <>')


IDENT = re.compile(r'[A-Za-z_$][\w$]*')
NUMLIT = re.compile(r'\d[\w]*')


def parse_primary(s, i):
    i = skip_ws(s, i)
    c = s[i]
    if c == '(':
        j = match_close(s, i) + 1
        k = skip_ws(s, j)
        if s.startswith('=>', k):
            return parse_type(s, k + 2)
        i = j
    elif c in '{[':
        i = match_close(s, i) + 1
    elif c in '"\'`':
        i = skip_string(s, i)
    elif NUMLIT.match(s, i):
        i = NUMLIT.match(s, i).end()
    else:
        m = IDENT.match(s, i)
        assert m, ('type expected', s[i:i + 60])
        i = m.end()
        if m.group(0) in ('typeof', 'keyof', 'readonly'):
            return parse_primary(s, i)
        while s[i:i + 1] == '.':
            i = IDENT.match(s, i + 1).end()
        if s[i:i + 1] == '<':
            i = match_angle(s, i) + 1
    while True:  # array postfix
        k = skip_ws(s, i)
        if s.startswith('[]', k):
            i = k + 2
        elif s[k:k + 1] == '[' and s[match_close(s, k) - 0] == ']' and re.fullmatch(r'\[\s*[\w\'"]*\s*\]', s[k:match_close(s, k) + 1]):
            i = match_close(s, k) + 1
        else:
            break
    return i


def parse_type(s, i):
    i = skip_ws(s, i)
    if s[i] in '|&':
        i += 1
    i = parse_primary(s, i)
    while True:
        k = skip_ws(s, i)
        if s[k:k + 1] in ('|', '&') and s[k + 1:k + 2] not in ('|', '&'):
            i = parse_primary(s, k + 1)
        else:
            return i


def split_params(body):
    parts, depth_start, i = [], 0, 0
    while i < len(body):
        c = body[i]
        if c in '"\'`':
            i = skip_string(body, i)
            continue
        if c in OPEN:
            i = match_close(body, i) + 1
            continue
        if c == '<' and re.search(r'[\w\]]$', body[:i].rstrip()) and ':' in body[depth_start:i]:
            try:
                i = match_angle(body, i) + 1
                continue
            except ValueError:
                pass
        if c == ',':
            parts.append(body[depth_start:i])
            depth_start = i + 1
        i += 1
    parts.append(body[depth_start:])
    return parts


def strip_param(p, props):
    """Remove modifiers / '?' / ': type' from one parameter; keep default value."""
    lead = re.match(r'\s*', p).group(0)
    q = p[len(lead):]
    is_prop = False
    while True:
        m = re.match(r'(public|private|protected|readonly)\s+', q)
        if not m:
            break
        is_prop = True
        q = q[m.end():]
    if not q.strip():
        return p
    if q[0] in '{[':
        j = match_close(q, 0) + 1
    elif q.startswith('...'):
        j = IDENT.match(q, 3).end()
    else:
        m = IDENT.match(q)
        if not m:
            return p
        j = m.end()
    name = q[:j]
    k = skip_ws(q, j)
    if q[k:k + 1] == '?':
        k = skip_ws(q, k + 1)
    if q[k:k + 1] == ':':
        k = parse_type(q, k + 1)
    rest = q[k:]
    if is_prop:
        props.append(name.strip())
    if rest.strip():
        return lead + name + ' ' + rest.strip()
    return lead + name


def strip_params(body, props=None):
    if props is None:
        props = []
    if not body.strip():
        return body
    return ','.join(strip_param(p, props) for p in split_params(body))


def remove_blocks(s):
    # interface X<...> { ... }
    while True:
        m = re.search(r'^(export\s+)?interface\s+\w+[^{]*\{', s, re.M)
        if not m:
            break
        e = match_close(s, m.end() - 1) + 1
        s = s[:m.start()] + s[e:]
    # top-level / nested "type X = ...;"
    while True:
        m = re.search(r'^[ \t]*(export\s+)?type\s+\w+(<[^=]*>)?\s*=', s, re.M)
        if not m:
            break
        e = parse_type(s, m.end())
        e = skip_ws(s, e)
        if s[e:e + 1] == ';':
            e += 1
        s = s[:m.start()] + s[e:]
    s = re.sub(r'^declare\s+[^;]*;\s*$', '', s, flags=re.M)
    return s


def strip_generics_after(s, pattern):
    out, pos = [], 0
    for m in re.finditer(pattern, s):
        if m.start() < pos:
            continue
        k = m.end()
        if s[k:k + 1] == '<':
            e = match_angle(s, k) + 1
            out.append(s[pos:k])
            pos = e
    out.append(s[pos:])
    return ''.join(out)


def fix_classes(s):
    s = re.sub(r'\babstract\s+class\b', 'class', s)
    s = strip_generics_after(s, r'\bclass\s+\w+')
    s = strip_generics_after(s, r'\bextends\s+\w+')
    s = re.sub(r'(class\s+\w+(?:\s+extends\s+\w+)?)\s+implements\s+[^{]*\{', r'\1 {', s)
    s = re.sub(r'\bstatic\s+readonly\b', 'static', s)
    # bare field declarations:  [private|public] [readonly] name[?]: Type;
    def field(m):
        return ''
    s = re.sub(r'^[ \t]+(?:(?:private|public|protected)\s+)?(?:readonly\s+)?[\w$]+\??\s*:\s*[^;=(){}]*;[ \t]*\n', field, s, flags=re.M)
    s = re.sub(r'^([ \t]+)(?:private|public|protected)\s+(?=(?:static\s+|async\s+)?[\w$\[])', r'\1', s, flags=re.M)
    return s


def fix_headers(s):
    """Strip parameter/return types from function, method and arrow headers; drop overload signatures."""
    out = []
    i = 0
    n = len(s)
    while i < n:
        c = s[i]
        if c in '"\'`':
            j = skip_string(s, i)
            out.append(s[i:j])
            i = j
            continue
        if c == '/' and s[i + 1:i + 2] == '/':
            j = s.index('\n', i)
            out.append(s[i:j])
            i = j
            continue
        if c == '/' and s[i + 1:i + 2] == '*':
            j = s.index('*/', i) + 2
            out.append(s[i:j])
            i = j
            continue
        if c == '(':
            close = match_close(s, i)
            after = skip_ws(s, close + 1)
            ret_end = None
            if s[after:after + 1] == ':':
                try:
                    te = parse_type(s, after + 1)
                    ta = skip_ws(s, te)
                    if s.startswith('=>', ta) or (ta < n and s[ta] in '{;'):
                        ret_end = te
                        after = ta
                except (AssertionError, ValueError, AttributeError):
                    ret_end = None
            prev = ''.join(out)[-200:]
            pm = re.search(r'([\w$\]>]+)\s*$', prev)
            prev_word = pm.group(1) if pm else ''
            is_arrow = s.startswith('=>', after)
            is_def = False
            if not is_arrow and after < n and s[after] in '{;' and prev_word and prev_word not in KEYWORDS:
                # method / function definition (or overload signature when followed by ';' with a return type)
                line_start = prev.rfind('\n') + 1
                head = prev[line_start:]
                if re.match(r'\s*(export\s+)?(static\s+)?(async\s+)?(function\s*[\w$]*|constructor|get\s+[\w$]+|[\w$]+|\[[^\]]*\])\s*$', head) and \
                        not re.match(r'\s*(if|for|while|switch|catch|return|throw|await)\b', head.strip() + ' '):
                    is_def = s[after] == '{' or ret_end is not None
            if is_arrow or is_def:
                if is_def and s[after] == ';':
                    # overload signature: remove the whole statement (back to line start)
                    text = ''.join(out)
                    ls = text.rfind('\n') + 1
                    out = [text[:ls]]
                    i = after + 1
                    continue
                props = []
                inner = strip_params(fix_headers(s[i + 1:close]), props)
                out.append('(' + inner + ')')
                if is_def and props:
                    # constructor parameter properties
                    assert s[after] == '{'
                    out.append(' {' + ''.join(' this.%s = %s;' % (p, p) for p in props))
                    i = after + 1
                    continue
                out.append(' ')
                i = after if ret_end is not None else close + 1
                continue
            # plain parenthesis: recurse into the content (may contain arrows)
            out.append('(' + fix_headers(s[i + 1:close]) + ')')
            i = close + 1
            continue
        out.append(c)
        i += 1
    return ''.join(out)


def strip_casts(s):
    s = re.sub(r'<any>', '', s)
    out, i = [], 0
    for m in re.finditer(r'\s+as\s+(?=[\[A-Za-z{])', s):
        if m.start() < i:
            continue
        # skip 'as' inside import/export lists
        line = s[s.rfind('\n', 0, m.start()) + 1:s.find('\n', m.end())]
        if re.match(r'\s*(import|export)\b', line):
            continue
        e = parse_type(s, m.end())
        out.append(s[i:m.start()])
        i = e
    out.append(s[i:])
    s = ''.join(out)
    s = re.sub(r'\)!(?=[;.\s)])', ')', s)
    return s


def strip_var_annotations(s):
    out, i = [], 0
    for m in re.finditer(r'\b(let|const|var)\s+([\w$]+)\s*:', s):
        if m.start() < i:
            continue
        e = parse_type(s, m.end())
        out.append(s[i:m.end() - 1].rstrip())
        out.append(' ')
        i = e
    out.append(s[i:])
    return ''.join(out)


def strip_fn_generics(s):
    s = strip_generics_after(s, r'\bfunction\s+[\w$]+')
    s = strip_generics_after(s, r'\bnew\s+[\w$.]+')
    # method generics:  name<TT extends this>(
    s = re.sub(r'^(\s+[\w$]+)<[^()\n]*>(?=\()', r'\1', s, flags=re.M)
    return s


def convert(src, is_index):
    s = src
    s = remove_blocks(s)
    s = fix_classes(s)
    s = strip_fn_generics(s)
    s = strip_casts(s)
    s = strip_var_annotations(s)
    s = fix_headers(s)
    s = s.replace("'./math.js'", "'./math.mjs'")
    if is_index:
        # class field without initialiser was deleted by fix_classes; nothing else to do
        pass
    return s


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
    out = sys.argv[2] if len(sys.argv) > 2 else '/tmp/nbls_ref'
    os.makedirs(out, exist_ok=True)
    for name in ('math', 'index'):
        with open(os.path.join(ref, name + '.ts')) as f:
            src = f.read()
        js = convert(src, name == 'index')
        with open(os.path.join(out, name + '.mjs'), 'w') as f:
            f.write(js)
    print('wrote', out)


if __name__ == '__main__':
    main()

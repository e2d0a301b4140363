"""Re-wraps the prose of a markdown file to a column limit (default 118): paragraphs and bullet items are refilled,
indented code blocks (4 spaces), headings and blank lines are left alone.  python tools/reflow_md.py DESIGN.md [width]"""
import re
import sys
import textwrap


def main():
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 118
    out, para = [], []

    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r'^(\s*)([*-] |\d+\. )?', first)
        indent, bullet = m.group(1), m.group(2) or ''
        text = ' '.join(p.strip() for p in para)
        text = text[len(bullet):] if bullet and text.startswith(bullet) else text
        sub = indent + ' ' * len(bullet)
        out.extend(textwrap.wrap(text, width=width, initial_indent=indent + bullet, subsequent_indent=sub,
                                 break_long_words=False, break_on_hyphens=False))
        para.clear()
    for line in open(path).read().split('\n'):
        if not line.strip() or line.startswith('#') or line.startswith('    ') or line.startswith('|') or line.startswith('```'):
            flush()
            out.append(line)
        elif re.match(r'^\s*([*-] |\d+\. )', line):
            flush()
            para.append(line)
        else:
            para.append(line)
    flush()
    open(path, 'w').write('\n'.join(out))


if __name__ == '__main__':
    main()

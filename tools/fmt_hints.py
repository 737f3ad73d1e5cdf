#!/usr/bin/env python3
"""pretty-print tools/time_igemm.py lines whose hint codes carry a tile_order (code = tile_order * 100 + hint)"""
import sys
for line in sys.stdin:
    if not line.startswith("idx"):
        continue
    parts = line.strip().split('|')
    print(parts[0])
    for q in parts[1:]:
        q = q.strip()
        if not q:
            continue
        h = int(q.split(':')[0][1:]); to = h // 100
        print('   hint', h % 100, 'order', to & 15, 'P', (to >> 4) & 15, 'K', to >> 8, q.split(':')[1])

"""The tail of ONE traced batch move-step by move-step (every 6th): live games, launches per 100 iterations, speculative rows per launch
(`DIEE_TRACE_STEPS=1 python3 bench.py --games 256 ... 2> trace.txt`; profiles/r05M_tail_by_phase_256_games.txt)."""
import sys,re
rows=[]
for l in open(sys.argv[1]):
    m=re.search(r'tail: (\d+) games, (\d+) iterations on (\d+) launches with rows \((\d+) pairs sent\), (\d+) speculative rows',l)
    if m: rows.append(tuple(map(int,m.groups())))
# one batch: move-steps in order; print every 8th
print('move-step | live games | launches per 100 iterations | spec rows per launch')
for i,(n,it,L,sent,S) in enumerate(rows):
    if i%6==0: print('%4d | %4d | %5.1f | %6.1f'%(i,n,100.0*L/it,S/max(L,1)))

"""The tail of a batch by live games: `DIEE_TRACE_STEPS=1 python3 bench.py ... 2> trace.txt; python3 tests/tools/tail_bands.py trace.txt` prints, per band of
live games, the move-steps, the search iterations, the network launches that carried rows, launches per iteration, iterations per launch and the
speculative rows per launch (profiles/r05B_*, r05G_*, r05H_*).  Reads the engine's trace lines only."""
import sys,re,collections
agg=collections.defaultdict(lambda:[0,0,0,0])
for l in open(sys.argv[1]):
    m=re.search(r'tail: (\d+) games, (\d+) iterations on (\d+) launches with rows \((\d+) pairs sent\), (\d+) speculative rows',l)
    if not m: continue
    n,it,L,sent,S=map(int,m.groups())
    b = (1,1) if n==1 else (2,2) if n==2 else (3,4) if n<=4 else (5,9) if n<=9 else (10,16) if n<=16 else (17,32) if n<=32 else (33,64) if n<=64 else (65,96)
    a=agg[b]; a[0]+=1; a[1]+=it; a[2]+=L; a[3]+=S
for b in sorted(agg):
    c,it,L,S=agg[b]; print('%3d ... %3d | %3d | %5d | %5d | %.3f | %4.1f | %5.1f'%(b[0],b[1],c,it,L,L/it,it/max(L,1),S/max(L,1)))

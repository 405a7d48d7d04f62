def sim(C, MODE, LATE, tiles=3, N=None, dW=0):
    NKS=C//16; PW=NKS//4; NBUF=4 if C==256 else 2
    N = N or (C*4)//32
    NL=2 if MODE==1 else 0; NS=2 if MODE==1 else 4
    q=PW+NS+NL; r=PW
    WQ = (NS+NL+(NBUF-1)*(q+r) if LATE else (NBUF-1)*q+(NBUF-2)*r) + dW
    WR = ((NBUF-1)*(q+r) if LATE else NS+NL+(NBUF-1)*r+(NBUF-2)*q) + dW
    queue=[]; bufA=[None]*NBUF; bufB=[None]*NBUF
    st=dict(aC=0,aB=0,bC=0,bB=0,arB=0,brB=0)
    cnt={}
    def complete(op):
        if op[0]=='X': return
        k=(op[0],op[2]); cnt[k]=cnt.get(k,0)+1
        if cnt[k]==PW:
            cnt[k]=0
            if op[0]=='A': bufA[op[2]]=op[1]
            else: bufB[op[2]]=op[1]
    def wait(n):
        while len(queue)>n: complete(queue.pop(0))
    def issueA():
        bufA[st['aB']]=('inflight',)
        for _ in range(PW): queue.append(('A',('A',st['aC']),st['aB']))
        st['aC']=(st['aC']+1)%N; st['aB']=(st['aB']+1)%NBUF
    def issueB():
        bufB[st['bB']]=('inflight',)
        for _ in range(PW): queue.append(('B',('B',st['bC']),st['bB']))
        st['bC']=(st['bC']+1)%N; st['bB']=(st['bB']+1)%NBUF
    snaps=[]
    def barrier(): snaps.append((list(bufA), list(bufB)))
    def vis(): return snaps[-1] if LATE else snaps[-2]
    def readA(expect):
        got=vis()[0][st['arB']]
        # WAR: the buffer must not have been re-issued before this read
        assert bufA[st['arB']]==('A',expect), ('overwritten', bufA[st['arB']], expect)
        st['arB']=(st['arB']+1)%NBUF
        assert got==('A',expect), (got, expect)
    def readB(expect):
        got=vis()[1][st['brB']]
        assert bufB[st['brB']]==('B',expect), ('overwritten', bufB[st['brB']], expect)
        st['brB']=(st['brB']+1)%NBUF
        assert got==('B',expect), (got, expect)
    for i in range(NBUF): issueA()
    for i in range(NBUF-1): issueB()
    wait(0); barrier()
    for t in range(tiles):
        queue.extend([('X',)]*NKS); wait(0); barrier()
        barrier(); readA(0)                  # P0
        for j in range(N):
            barrier()                        # Q(j)
            if j==0: issueA()
            issueB()
            if j+1<N: readA(j+1)
            queue.extend([('X',)]*(NS+NL))
            wait(WQ)
            barrier()                        # R(j)
            if j<=N-2: issueA()
            readB(j)
            wait(WR)
        queue.extend([('X',)]*(C//16))
    return 'ok'
for C in (256,512):
    PWX=C//64
    for MODE in (0,1):
        for LATE in (False, True):
            res=[]
            for dW in (0,1,PWX):
                try: res.append(sim(C, MODE, LATE, dW=dW))
                except AssertionError as e: res.append('FAIL %s'%(e,))
            print(C, MODE, LATE, res)

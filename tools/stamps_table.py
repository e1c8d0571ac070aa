import sys
rows=[l.split() for l in sys.stdin if 'stamps-wg' in l]
n=int(sys.argv[1])
rows=rows[-n:]
wg=[(int(r[1]),int(r[2])//1000,int(r[3]),int(r[4])) for r in rows]
tiles=[];cur=[]
for w in wg:
    if w[2]==0 and cur: tiles.append(cur);cur=[]
    cur.append(w)
tiles.append(cur)
nb=8;k=0
for I in range(nb):
    for J in range(I,nb):
        t=tiles[k];k+=1
        print(I,J,' '.join('%4d'%x[1] for x in t))

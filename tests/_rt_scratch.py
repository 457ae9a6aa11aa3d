import sys; sys.path.insert(0,'/root/repo')
import numpy as np, streamgen, oracle
def rt(**kw):
    s, rec, sizes = streamgen.encode(**kw)
    try:
        out, info = oracle.decode(s, crop=False)
    except Exception as ex:
        print(kw, "DECODE FAIL", ex, len(s)); return False
    ok = out.shape==rec.shape and np.array_equal(out, rec)
    msg = ""
    if not ok and out.shape==rec.shape:
        W=(kw['width']+15)//16*16; H=(kw['height']+15)//16*16
        for f in range(out.shape[0]):
            if not np.array_equal(out[f],rec[f]):
                idx=np.nonzero(out[f]!=rec[f])[0]
                i=idx[0]
                if i<W*H: pos=('Y',i%W,i//W, 'mb',(i%W)//16,(i//W)//16)
                else:
                    j=(i-W*H)%(W*H//4); pos=('C',j%(W//2),j//(W//2),'mb',(j%(W//2))//8,(j//(W//2))//8)
                msg=f"first mismatch frame {f} at {pos} count {len(idx)} maxdiff {np.abs(out[f].astype(int)-rec[f]).max()}"; break
    print({k:v for k,v in kw.items() if k not in('width','height')}, len(s), "OK" if ok else "MISMATCH "+msg+str(out.shape)+str(rec.shape))
    return ok
if __name__=="__main__":
    rt(width=64,height=48,frames=2,profile_idc=66,cabac=0,idr_period=1)
    rt(width=64,height=48,frames=2,profile_idc=77,cabac=1,idr_period=1)
    rt(width=64,height=48,frames=4,profile_idc=66,cabac=0,idr_period=0)
    rt(width=64,height=48,frames=4,profile_idc=77,cabac=1,idr_period=0)

def matrix():
    ok=True
    base=dict(width=176,height=144,frames=5,idr_period=0)
    ok&=rt(**base,profile_idc=66,cabac=0,qp=20)
    ok&=rt(**base,profile_idc=77,cabac=1,qp=20)
    ok&=rt(**base,profile_idc=66,cabac=0,qp=8,noise=30)
    ok&=rt(**base,profile_idc=77,cabac=1,qp=8,noise=30)
    ok&=rt(**base,profile_idc=66,cabac=0,qp=40)
    ok&=rt(**base,profile_idc=77,cabac=1,qp=45)
    ok&=rt(**base,profile_idc=100,cabac=1,transform8x8=1,qp=26)
    ok&=rt(**base,profile_idc=100,cabac=0,transform8x8=1,qp=26)
    ok&=rt(**base,profile_idc=100,cabac=1,transform8x8=1,qp=26,scaling_matrix=1)
    ok&=rt(**base,profile_idc=77,cabac=1,slices=3,cabac_init_idc=-1)
    ok&=rt(**base,profile_idc=66,cabac=0,slices=4,deblock_idc=2)
    ok&=rt(**base,profile_idc=77,cabac=1,num_ref_frames=3,qp=30)
    ok&=rt(**base,profile_idc=66,cabac=0,num_ref_frames=4,qp=30)
    ok&=rt(**base,profile_idc=77,cabac=1,pcm_permille=60,qp_jitter=6)
    ok&=rt(**base,profile_idc=66,cabac=0,pcm_permille=60,qp_jitter=6)
    ok&=rt(**base,profile_idc=77,cabac=1,weighted_pred=1,num_ref_frames=2)
    ok&=rt(**base,profile_idc=77,cabac=1,constrained_intra=1,intra_in_p_permille=300)
    ok&=rt(**base,profile_idc=66,cabac=0,constrained_intra=1,intra_in_p_permille=300)
    ok&=rt(**base,profile_idc=77,cabac=1,deblock_idc=1)
    ok&=rt(**base,profile_idc=77,cabac=1,alpha_off_div2=3,beta_off_div2=-2,chroma_qp_offset=4)
    ok&=rt(**base,profile_idc=77,cabac=1,sub8x8_permille=600,skip_permille=100,qp=24)
    ok&=rt(**base,profile_idc=66,cabac=0,sub8x8_permille=600,skip_permille=100,qp=24,poc_type=2)
    ok&=rt(width=180,height=100,frames=3,idr_period=0,profile_idc=100,cabac=1,transform8x8=1,long_start_code=0,slices=2)
    ok&=rt(**base,profile_idc=77,cabac=1,cabac_init_idc=1)
    ok&=rt(**base,profile_idc=77,cabac=1,cabac_init_idc=2)
    print("ALL OK" if ok else "SOME FAILED")
if __name__=="__main__": matrix()

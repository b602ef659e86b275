import sys
sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/climaocean.jl_amd')
import numpy as np
import oracle as orc, numpy_oracle as npo
from coflux import synthetic as syn, interface_computations as ic
nx,ny,hx,hy=360,140,3,3
g=orc.make_grid(nx,ny,hx,hy,1)
oc=syn.ocean_state(nx,ny,hx,hy,ny_global=560,j_offset=200)
src=syn.jra55_snapshots(2)
fi,fj,phi=syn.latlon_fractional_indices(nx,ny,hx,hy,ny_global=560,j_offset=200)
at=orc.interpolate_atmosphere_state(g,src,dict(separable=True,fi=fi,fj=fj))
def setup(fluxes):
    th=npo.Thermo(ic.AtmosphereThermodynamicsParameters()); sw=ic.SeawaterComposition(); op=ic.OceanProperties()
    W=(slice(hy,hy+ny),slice(hx,hx+nx)); E=(slice(hy,hy+ny),slice(hx+1,hx+nx+1)); N=(slice(hy+1,hy+ny+1),slice(hx,hx+nx))
    uo=0.5*(oc['u'][W]+oc['u'][E]); vo=0.5*(oc['v'][W]+oc['v'][N]); Ts=oc['T'][W]+273.15; So=oc['S'][W]
    wet=oc['mask'][W]!=0
    ua,va,Ta,pa,qa=(at[k][W] for k in 'uvTpq')
    A=th.state_pTq(pa,Ta,qa); qs=npo.water_mole_fraction(sw,So)*th.svp_liquid(Ts)/(A['rho']*th.Rv*Ts)
    dq=th.q_vapor(A)-qs; dth=Ta+9.81*10/th.cp_m(A)-Ts; du,dv=ua-uo,va-vo
    S=th.state_pTq(pa,Ts,qs); Tv,qv=th.T_virtual(S),th.q_vapor(S); delta=th.eps-1
    dU=np.sqrt(du*du+dv*dv); kap=0.4; stab=fluxes.stability_functions.name
    coare=isinstance(fluxes.similarity_form, ic.COARELogarithmicSimilarityProfile)
    def F(x):
        us,ts,qq=x
        b=9.81/Tv*(ts*(1+delta*qv)+delta*Tv*qq); Jb=-us*b
        Ug=np.maximum(fluxes.gustiness_parameter*np.cbrt(np.maximum(Jb,0)*600),fluxes.minimum_gustiness)
        U=np.sqrt(du*du+dv*dv+Ug*Ug)
        lu=npo.momentum_length(fluxes.momentum_roughness_length,9.81,us,dU,Ts)
        lq=npo.scalar_length(fluxes.water_vapor_roughness_length,lu,us,Ts)
        with np.errstate(all='ignore'):
            L=np.where(b==0,np.inf,-us*us/(kap*b))
            def prof(psi,l):
                r=np.log(10/l)-psi(stab,10/L)
                r=r if coare else r+psi(stab,l/L)
                return np.maximum(r,1.0)
            return np.array([kap/prof(npo.psi_m,lu)*U, kap/prof(npo.psi_h,lq)*dth, kap/prof(npo.psi_h,lq)*dq])
    return F,wet
def run(name,fluxes):
    F,wet=setup(fluxes)
    x0=np.full((3,ny,nx),1e-4)
    # plain
    x=x0.copy(); n_plain=np.zeros((ny,nx),int); act=wet.copy()
    for it in range(100):
        xn=F(x); d=np.abs(xn-x).sum(0); x=np.where(act,xn,x); n_plain+=act; act=act&~(d<1e-8)
        if not act.any(): break
    xref=x.copy()
    for _ in range(30): xref=F(xref)   # true fixed point
    err_plain=np.abs(x-xref)/np.maximum(np.abs(xref),1e-12)
    print(name,'plain iters mean %.2f max %d; rel err of stopped iterate vs fixed point: max'%(n_plain[wet].mean(),n_plain.max()), err_plain[:,wet].max(1))
    # Anderson m, started after `warm` plain iterations
    for m in (1,2):
      for warm in (1,2,3):
        x=x0.copy(); hist_x=[];hist_f=[]; n=np.zeros((ny,nx),int); act=wet.copy(); done_x=x.copy()
        for it in range(60):
            gx=F(x); f=gx-x
            d=np.abs(f).sum(0)
            newly=act&(d<1e-8)
            done_x=np.where(newly,gx,done_x)
            n+=act; act=act&~newly
            if not act.any(): break
            hist_x.append(gx);hist_f.append(f)
            mk=min(m,len(hist_f)-1) if it>=warm else 0
            if mk==0: xn=gx
            elif mk==1:
                df=hist_f[-1]-hist_f[-2]; dg=hist_x[-1]-hist_x[-2]
                den=(df*df).sum(0); gam=np.where(den>0,(hist_f[-1]*df).sum(0)/np.where(den>0,den,1),0)
                xn=gx-gam*dg
            else:
                df1=hist_f[-1]-hist_f[-2]; df2=hist_f[-2]-hist_f[-3]; dg1=hist_x[-1]-hist_x[-2]; dg2=hist_x[-2]-hist_x[-3]
                a11=(df1*df1).sum(0);a12=(df1*df2).sum(0);a22=(df2*df2).sum(0); b1=(df1*hist_f[-1]).sum(0); b2=(df2*hist_f[-1]).sum(0)
                det=a11*a22-a12*a12; ok=det>1e-30*(a11*a22+1e-300)
                dets=np.where(ok,det,1)
                g1=np.where(ok,(b1*a22-b2*a12)/dets,np.where(a11>0,b1/np.where(a11>0,a11,1),0)); g2=np.where(ok,(a11*b2-a12*b1)/dets,0)
                xn=gx-g1*dg1-g2*dg2
            # safeguard: keep u* positive
            bad=(xn[0]<=0)|~np.isfinite(xn).all(0)
            xn=np.where(bad,gx,xn)
            x=np.where(act,xn,x)
        err=np.abs(done_x-xref)/np.maximum(np.abs(xref),1e-12)
        print('   AA(%d) warm %d: evals mean %.2f max %d  unconverged %d; rel err max'%(m,warm,n[wet].mean(),n.max(),act.sum()),err[:,wet].max(1))
run('default',ic.SimilarityTheoryFluxes())
run('corrected',ic.corrected_atmosphere_ocean_fluxes())

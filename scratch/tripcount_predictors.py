import sys
sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/climaocean.jl_amd')
import numpy as np
import oracle as orc
from coflux import synthetic as syn, interface_computations as ic
nx,ny,hx,hy=720,280,3,3
g=orc.make_grid(nx,ny,hx,hy,0)
oc=syn.ocean_state(nx,ny,hx,hy,ny_global=560,j_offset=140)
src=syn.jra55_snapshots(2)
fi,fj,phi=syn.latlon_fractional_indices(nx,ny,hx,hy,ny_global=560,j_offset=140)
at=orc.interpolate_atmosphere_state(g,src,dict(separable=True,fi=fi,fj=fj))
for name,f in (('default',ic.SimilarityTheoryFluxes()),('corrected',ic.corrected_atmosphere_ocean_fluxes())):
    P=ic.flux_params(f)
    a=orc.compute_atmosphere_ocean_fluxes(g,P,oc,at,nthreads=8)
    W=(slice(hy,hy+ny),slice(hx,hx+nx))
    wet=oc['mask'][W]!=0
    it=a['iterations'][W][wet].astype(float)
    U=np.hypot(at['u'][W],at['v'][W])[wet]
    dT=(at['T'][W]-273.15-oc['T'][W])[wet]
    us=a['friction_velocity'][W][wet]; ts=a['temperature_scale'][W][wet]; qs=a['humidity_scale'][W][wet]
    zeta=-10*0.4*(9.81/285*(ts+0.61*285*qs))/us**2
    print(name,'mean it %.2f'%it.mean(), 'corr with U %.2f, log U %.2f, dT %.2f, zeta %.2f, |zeta| %.2f log|zeta| %.2f'%(np.corrcoef(it,U)[0,1],np.corrcoef(it,np.log(U+0.1))[0,1],np.corrcoef(it,dT)[0,1],np.corrcoef(it,zeta)[0,1],np.corrcoef(it,np.abs(zeta))[0,1],np.corrcoef(it,np.log(np.abs(zeta)+1e-6))[0,1]))
    # wave-max if sorted by log U within chunks of 512
    n=(len(it)//512)*512
    itc=it[:n].reshape(-1,512); Uc=U[:n].reshape(-1,512)
    base=itc.reshape(-1,8,64).max(2).mean()
    order=np.argsort(Uc,axis=1); its=np.take_along_axis(itc,order,1)
    srt=its.reshape(-1,8,64).max(2).mean()
    order=np.argsort(itc,axis=1); its=np.take_along_axis(itc,order,1)
    perfect=its.reshape(-1,8,64).max(2).mean()
    print('   wave-max mean: unsorted %.2f  sorted-by-U %.2f  sorted-by-true-count %.2f  (mean count %.2f)'%(base,srt,perfect,it.mean()))
print('--- regression predictors')
for name,f in (('default',ic.SimilarityTheoryFluxes()),('corrected',ic.corrected_atmosphere_ocean_fluxes())):
    P=ic.flux_params(f)
    a=orc.compute_atmosphere_ocean_fluxes(g,P,oc,at,nthreads=8)
    W=(slice(hy,hy+ny),slice(hx,hx+nx)); E=(slice(hy,hy+ny),slice(hx+1,hx+nx+1)); N=(slice(hy+1,hy+ny+1),slice(hx,hx+nx))
    wet=oc['mask'][W]!=0
    it=a['iterations'][W][wet].astype(float)
    uo=0.5*(oc['u'][W]+oc['u'][E]); vo=0.5*(oc['v'][W]+oc['v'][N])
    dU=np.hypot(at['u'][W]-uo,at['v'][W]-vo)[wet]
    dT=(at['T'][W]-273.15-oc['T'][W])[wet]
    lU=np.log(dU+0.2)
    feats={'logU':[lU],'logU,dT':[lU,dT],'logU,dT,dT/U2':[lU,dT,dT/(dU**2+0.04)],'quad':[lU,dT,lU*lU,dT*dT,lU*dT,dT/(dU**2+0.04)]}
    n=(len(it)//512)*512
    for fn,fs in feats.items():
        X=np.stack([np.ones_like(it)]+fs,1); beta,*_=np.linalg.lstsq(X,it,rcond=None); pred=X@beta
        for ch in (512,2048):
            n=(len(it)//ch)*ch
            itc=it[:n].reshape(-1,ch); pc=pred[:n].reshape(-1,ch)
            order=np.argsort(pc,axis=1); its=np.take_along_axis(itc,order,1)
            print(name,fn,'chunk',ch,'wave-max mean %.2f'%its.reshape(-1,ch//64,64).max(2).mean(),'corr %.2f'%np.corrcoef(pred,it)[0,1])

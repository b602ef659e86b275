import sys
sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/climaocean.jl_amd')
import numpy as np
import oracle as orc
from coflux import synthetic as syn, interface_computations as ic
nx,ny,hx,hy=1440,560,3,3
g=orc.make_grid(nx,ny,hx,hy,0)
oc=syn.ocean_state(nx,ny,hx,hy)
src=syn.jra55_snapshots(2)
fi,fj,phi=syn.latlon_fractional_indices(nx,ny,hx,hy)
at=orc.interpolate_atmosphere_state(g,src,dict(separable=True,fi=fi,fj=fj))
for name,mk in (('default',lambda: ic.SimilarityTheoryFluxes()),('corrected',ic.corrected_atmosphere_ocean_fluxes)):
    f=mk(); P=ic.flux_params(f)
    a=orc.compute_atmosphere_ocean_fluxes(g,P,oc,at,nthreads=8)
    f2=mk(); f2.solver_stop_criteria=ic.ConvergenceStopCriteria(1e-15,5000); P2=ic.flux_params(f2)
    b=orc.compute_atmosphere_ocean_fluxes(g,P2,oc,at,nthreads=8)
    W=(slice(hy,hy+ny),slice(hx,hx+nx))
    wet=oc['mask'][W]!=0
    print(name,'iters tol1e-8: mean %.2f max %d | tol1e-15: mean %.1f max %d'%(a['iterations'][W][wet].mean(),a['iterations'].max(),b['iterations'][W][wet].mean(),b['iterations'].max()))
    for k,sc in (('sensible_heat',1.0),('latent_heat',1.0),('water_vapor',1e-6),('x_momentum',1e-3),('y_momentum',1e-3),('friction_velocity',1e-3)):
        e=np.abs(a[k][W]-b[k][W])/np.maximum(np.abs(b[k][W]),sc)
        e=e[wet]
        print('   %-18s max %.2e  frac>1e-6: %.2e  frac>1e-7: %.2e frac>1e-8 %.2e'%(k,e.max(),(e>1e-6).mean(),(e>1e-7).mean(),(e>1e-8).mean()))
    e=np.abs(a['friction_velocity'][W]-b['friction_velocity'][W])/np.maximum(np.abs(b['friction_velocity'][W]),1e-3)
    e=np.where(wet,e,0); idx=np.argsort(e.ravel())[-5:]
    for t in idx:
        j,i=divmod(t,nx); k=(j+hy,i+hx)
        print('     cell',j,i,'err %.2e'%e[j,i],'it',a['iterations'][k],b['iterations'][k],'u* %.5f th* %.5f q* %.3e'%(b['friction_velocity'][k],b['temperature_scale'][k],b['humidity_scale'][k]),'Ta-To %.2f'%(at['T'][k]-273.15-oc['T'][k]),'|ua| %.2f'%np.hypot(at['u'][k],at['v'][k]))

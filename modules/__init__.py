"""Import-path shim: the reference's notebooks do `from modules.ddpm_tasks import ddpm_run`,
`from modules.ddpm_models import *` ... (Train.ipynb:23, Results.ipynb:33-39).  These modules
re-export the MI355X engine (`afdm`) under those names so the notebooks run unmodified."""

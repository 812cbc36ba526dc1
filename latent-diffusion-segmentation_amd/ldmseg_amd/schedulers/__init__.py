from .ddim_scheduler import DDIMNoiseScheduler, DDIMNoiseSchedulerOutput  # noqa: F401

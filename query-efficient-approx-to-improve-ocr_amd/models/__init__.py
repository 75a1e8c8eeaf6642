"""Network containers of the hot path (UNet cleaner, CRNN proxy); forward passes run the HIP kernel schedules in qea/."""

from accflow_amd.networks.AccFlow_ import (AccFlow, AccPlus, Blending, FlowDecoder, FlowEncoder,  # noqa: F401
                                            downflow8, getOcc)
